mkdir -p gpurun_out/r05 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r05/smoke:300:python3 -c 'import __graft_entry__ as g; g.smoke()'" \
 "r05/collect_profiles:900:bash tools/collect_profiles.sh r05" \
 "r05/collect_config_profiles:1000:bash tools/collect_config_profiles.sh r05" \
 "r05/bench_configs:600:python3 tools/bench_configs.py" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r05/session6_summary.txt
