mkdir -p gpurun_out/r05 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r05/fuzz_big:900:python3 tools/fuzz_gpu_vs_oracle.py 200000 23" \
 "r05/gputests_ab_full:1100:ANEMOI_MI355X_LIB=$GRAFT_REPO_ROOT/anemoi-rust_amd/lib/libanemoi_ab.so python3 -m pytest tests -m gpu -x -q --durations=4" \
 "r05/jive_queue_ab:300:ANEMOI_MI355X_LIB=$GRAFT_REPO_ROOT/anemoi-rust_amd/lib/libanemoi_ab.so python3 tools/exp_jive_queue.py" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r05/session10_summary.txt
