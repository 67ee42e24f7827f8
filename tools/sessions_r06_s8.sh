mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/cfg3_balance_on:300:python3 tools/exp_cfg3_after_other_kernels.py" \
 "r06/cfg3_balance_off:300:python3 tools/exp_cfg3_after_other_kernels.py no_balance" \
 "r06/bench_configs:900:python3 tools/bench_configs.py" \
 "r06/gpu_suite_s8:1100:python3 -m pytest tests -m gpu -q" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session8_summary.txt
