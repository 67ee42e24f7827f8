mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/sampler_queue_collision:400:python3 tools/exp_sampler_queue_collision.py 24" \
 "r06/cycles_and_probe_tests:600:python3 -m pytest tests/test_gpu_cycles.py tests/test_gpu_configs.py -m gpu -q -k \"cycles or warmup_and_issue\"" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session13_summary.txt
