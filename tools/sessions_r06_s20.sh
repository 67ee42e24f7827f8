mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/sampler_priority_placement:400:python3 tools/exp_sampler_priority_placement.py" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session20_summary.txt
