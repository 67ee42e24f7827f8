mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/lane_priorities:600:python3 tools/exp_lane_priorities.py" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session15_summary.txt
