mkdir -p gpurun_out/r05 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r05/cfg3_plain_d:200:python3 tools/exp_cfg3_repeat.py" \
 "r05/cfg3_api_warmup_d:200:python3 tools/exp_cfg3_repeat.py api_warmup" \
 "r05/gputests_s4:1100:python3 -m pytest tests -m gpu -x -q --durations=12" \
 "r05/seg_latency_product:300:python3 tools/bench_segmented_latency.py" \
 "r05/seg_latency_r04:300:ANEMOI_MI355X_LIB=$GRAFT_REPO_ROOT/anemoi-rust_amd/lib/libanemoi_ab_r04.so python3 tools/bench_segmented_latency.py" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r05/session4_summary.txt
