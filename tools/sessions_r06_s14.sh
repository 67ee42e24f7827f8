mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/gpu_suite_reverse2:1100:ANEMOI_TEST_ORDER=reverse python3 -m pytest tests -m gpu -q" \
 "r06/gpu_suite_shuffle7:1100:ANEMOI_TEST_ORDER=shuffle:7 python3 -m pytest tests -m gpu -q" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session14_summary.txt
