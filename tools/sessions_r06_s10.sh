mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/placement_test:400:python3 -m pytest tests/test_gpu_configs.py -m gpu -q -s -k underfilled_launch_is_placed" \
 "r06/first_result:300:python3 tools/first_result.py anemoi-rust_amd/lib/libanemoi_ab.so anemoi-rust_amd/lib/libanemoi_mi355x.so anemoi-rust_amd/lib/libanemoi_ab.so anemoi-rust_amd/lib/libanemoi_mi355x.so" \
 "r06/gpu_suite_product:1100:python3 -m pytest tests -m gpu -q --durations=6" \
 "r06/gpu_suite_ab:1100:ANEMOI_MI355X_LIB=$GRAFT_REPO_ROOT/anemoi-rust_amd/lib/libanemoi_ab.so python3 -m pytest tests -m gpu -q" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session10_summary.txt
