mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/new_queue_placement:300:python3 tools/exp_new_queue_placement.py" \
 "r06/xcd_accounting2:300:ANEMOI_MI355X_LIB=anemoi-rust_amd/lib/libanemoi_ab.so python3 tools/exp_xcd_accounting.py --rounds 3" \
 "r06/collect_profiles:1000:bash tools/collect_profiles.sh r06" \
 "r06/collect_config_profiles:1000:bash tools/collect_config_profiles.sh r06" \
 "r06/cycles_budget:400:python3 tools/measure_cycles.py --reps 5 --best --out gpurun_out/r06/cycles_budget.json" \
 "r06/cycles_tests:600:python3 -m pytest tests/test_gpu_cycles.py -m gpu -q -s" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session4_summary.txt; cp -r profiles/r06 gpurun_out/r06/profiles_r06
