mkdir -p gpurun_out/r05 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
PMC="GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" && \
tools/gpu_session.sh \
 "r05/pmc_cfg3_plain3:300:rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/r05/pmc_cfg3_plain3 -o run -- python3 tools/exp_cfg3_repeat.py" \
 "r05/collect_profiles:900:bash tools/collect_profiles.sh r05" \
 "r05/bench_configs:600:python3 tools/bench_configs.py" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r05/session5_summary.txt; mkdir -p gpurun_out/r05/prof_r05_copy && cp gpurun_out/prof_r05/*.json gpurun_out/prof_r05/*.csv gpurun_out/r05/prof_r05_copy/ 2>/dev/null; ls gpurun_out/prof_r05 | head -30
