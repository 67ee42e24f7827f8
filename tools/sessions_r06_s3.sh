mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/gpu_suite_s3:1100:python3 -m pytest tests -m gpu -q" \
 "r06/xcd_accounting:500:ANEMOI_MI355X_LIB=anemoi-rust_amd/lib/libanemoi_ab.so python3 tools/exp_xcd_accounting.py --rounds 3" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session3_summary.txt
