mkdir -p gpurun_out/r05 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r05/cfg3_plain_c:200:python3 tools/exp_cfg3_repeat.py" \
 "r05/cfg3_api_warmup:200:python3 tools/exp_cfg3_repeat.py api_warmup" \
 "r05/place_plain4:120:tools/ubench/bin/first_launch_placement" \
 "r05/gputests_s3:1100:python3 -m pytest tests -m gpu -x -q --durations=12" \
 "r05/bench_n1:300:python3 bench.py" \
 "r05/seg_latency_product:300:python3 tools/bench_segmented_latency.py" \
 "r05/seg_latency_r04:300:ANEMOI_MI355X_LIB=$GRAFT_REPO_ROOT/anemoi-rust_amd/lib/libanemoi_ab_r04.so python3 tools/bench_segmented_latency.py" \
 "r05/host_api_ragged:600:python3 tools/bench_host_api.py ragged" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r05/session3_summary.txt
