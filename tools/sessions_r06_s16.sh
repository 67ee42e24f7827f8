mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/generic_unreduced:600:python3 -m pytest tests/test_gpu_generic.py -m gpu -q -k unreduced" \
 "r06/gpu_suite_s16:1100:python3 -m pytest tests -m gpu -q" \
 "r06/bench_default:400:python3 bench.py > gpurun_out/r06/bench_default_s16.json" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session16_summary.txt
