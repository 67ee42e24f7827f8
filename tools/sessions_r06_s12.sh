mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/capture_tests:400:python3 -m pytest tests/test_gpu_capture.py -m gpu -q" \
 "r06/merkle_levels:400:python3 tools/bench_merkle.py jubjub 21" \
 "r06/fuzz_big:900:python3 tools/fuzz_gpu_vs_oracle.py 200000 23" \
 "r06/gpu_suite_reverse:1100:ANEMOI_TEST_ORDER=reverse python3 -m pytest tests -m gpu -q" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session12_summary.txt
