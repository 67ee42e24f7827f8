mkdir -p gpurun_out/r05 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
PMC="GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" && \
tools/gpu_session.sh \
 "r05/first_result:200:python3 tools/first_result.py anemoi-rust_amd/lib/libanemoi_ab.so anemoi-rust_amd/lib/libanemoi_mi355x.so" \
 "r05/gputests_product:1000:python3 -m pytest tests -m gpu -x -q --durations=15" \
 "r05/gputests_ab_negatives:600:ANEMOI_MI355X_LIB=$GRAFT_REPO_ROOT/anemoi-rust_amd/lib/libanemoi_ab.so python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k 'cooperative_and_lane_private or fuzz'" \
 "r05/cfg3_plain_a:200:python3 tools/exp_cfg3_repeat.py" \
 "r05/cfg3_plain_b:200:python3 tools/exp_cfg3_repeat.py" \
 "r05/pmc_cfg3_plain2:300:rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/r05/pmc_cfg3_plain2 -o run -- python3 tools/exp_cfg3_repeat.py" \
 "r05/place_plain2:120:tools/ubench/bin/first_launch_placement" \
 "r05/place_plain3:120:tools/ubench/bin/first_launch_placement grid=4096" \
 "r05/ab_r03_r04_swapped:300:python3 tools/ab_bench.py --rounds 9 r04=anemoi-rust_amd/lib/libanemoi_ab.so r03=anemoi-rust_amd/lib/libanemoi_r03.so product=anemoi-rust_amd/lib/libanemoi_mi355x.so" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r05/session2_summary.txt
