mkdir -p gpurun_out/r05 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r05/collect_profiles:900:bash tools/collect_profiles.sh r05" \
 "r05/gputests_final:1100:python3 -m pytest tests -m gpu -x -q --durations=8" \
 "r05/host_api:1000:python3 tools/bench_host_api.py" \
 "r05/fuzz:600:python3 tools/fuzz_gpu_vs_oracle.py 20000 11" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r05/session7_summary.txt
