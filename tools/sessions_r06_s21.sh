mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/cycles_fixed_1:300:python3 tools/measure_cycles.py --reps 5 --best --out gpurun_out/r06/cycles_budget.json" \
 "r06/cycles_fixed_2:300:python3 tools/measure_cycles.py --reps 5 --best" \
 "r06/sampler_priority_placement_reuse:400:python3 tools/exp_sampler_priority_placement.py reuse" \
 "r06/sampler_queue_collision2:400:python3 tools/exp_sampler_queue_collision.py 16" \
 "r06/bench_default:400:python3 bench.py > gpurun_out/r06/bench_default_s21.json" \
 "r06/cycles_tests:600:python3 -m pytest tests/test_gpu_cycles.py tests/test_gpu_configs.py -m gpu -q -k 'cycles or warmup_and_issue'" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session21_summary.txt
