mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/cfg3_spread:300:python3 tools/exp_cfg3_after_other_kernels.py" \
 "r06/cfg3_no_spread:300:python3 tools/exp_cfg3_after_other_kernels.py no_spread" \
 "r06/smoke_after_spread:300:python3 __graft_entry__.py smoke" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session6_summary.txt
