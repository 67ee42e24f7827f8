mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/gpu_session.sh \
 "r06/cfg3_sampler_high:300:python3 tools/exp_cfg3_after_other_kernels.py" \
 "r06/cfg3_sampler_default:300:python3 tools/exp_cfg3_after_other_kernels.py sampler_default_priority" \
 ; cp gpurun_out/session_summary.txt gpurun_out/r06/session18_summary.txt
