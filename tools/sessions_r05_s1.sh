mkdir -p gpurun_out/r05 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
PMC="GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" && \
tools/gpu_session.sh \
 "r05/place_plain:120:tools/ubench/bin/first_launch_placement" \
 "r05/place_warm2:120:tools/ubench/bin/first_launch_placement warm=2" \
 "r05/place_warm128:120:tools/ubench/bin/first_launch_placement warm=128" \
 "r05/place_idle:120:tools/ubench/bin/first_launch_placement idle_ms=400" \
 "r05/cfg3_plain_notool:200:python3 tools/exp_cfg3_repeat.py" \
 "r05/pmc_cfg3_plain:300:rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/r05/pmc_cfg3_plain -o run -- python3 tools/exp_cfg3_repeat.py" \
 "r05/pmc_cfg3_warm:300:rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/r05/pmc_cfg3_warm -o run -- python3 tools/exp_cfg3_repeat.py warm=4096" \
 "r05/ab_r03_r04:300:python3 tools/ab_bench.py --rounds 9 r03=anemoi-rust_amd/lib/libanemoi_r03.so r04=anemoi-rust_amd/lib/libanemoi_mi355x.so" \
 "r05/pmc_trees:300:rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/r05/pmc_trees -o run -- python3 tools/profile_workloads.py cfg5 --reps 3" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r05/session1_summary.txt
