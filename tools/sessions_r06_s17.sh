mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/collect_profiles:1000:bash tools/collect_profiles.sh r06" \
 "r06/collect_config_profiles:1000:bash tools/collect_config_profiles.sh r06" \
 "r06/cycles_budget:400:python3 tools/measure_cycles.py --reps 5 --best --out gpurun_out/r06/cycles_budget.json" \
 "r06/bench_driver_command:400:python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_n1_driver_command.json" \
 "r06/merkle_balance:400:python3 tools/exp_merkle_balance.py" \
 "r06/bench_configs:900:python3 tools/bench_configs.py" \
 "r06/gpu_suite_final:1100:python3 -m pytest tests -m gpu -q" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session17_summary.txt
