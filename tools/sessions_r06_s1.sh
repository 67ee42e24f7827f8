mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/rccl_one_rank:600:python3 -m pytest tests/test_gpu_configs.py -m gpu -x -q -k one_rank_over_rccl -s" \
 "r06/cycles_tool:400:python3 tools/measure_cycles.py --reps 3 --out gpurun_out/r06/cycles_s1.json" \
 "r06/cycles_tests:600:python3 -m pytest tests/test_gpu_cycles.py -m gpu -q -s" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session1_summary.txt
