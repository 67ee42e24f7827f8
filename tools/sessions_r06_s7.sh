mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/balance_launch2:400:ANEMOI_MI355X_LIB=anemoi-rust_amd/lib/libanemoi_ab.so python3 tools/exp_balance_launch2.py" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session7_summary.txt
