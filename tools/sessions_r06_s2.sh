mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/malformed_offsets:600:python3 -m pytest tests/test_gpu_configs.py -m gpu -x -q -k malformed_device_offsets -s" \
 "r06/unreduced_inputs:900:python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k unreduced_abi_inputs" \
 "r06/gpu_suite_s2:1100:python3 -m pytest tests -m gpu -x -q" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session2_summary.txt
