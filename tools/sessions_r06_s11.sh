mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/host_api:1000:python3 tools/bench_host_api.py" \
 "r06/ragged_small:600:python3 tools/bench_ragged_small.py" \
 "r06/merkle_levels:400:python3 tools/bench_merkle.py" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session11_summary.txt
