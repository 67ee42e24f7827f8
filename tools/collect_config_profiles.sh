#!/bin/bash
# Counter evidence for the kernels that are NOT the headline (config 3: k_sponge_pair<2, true>; config 5:
# k_jive<4, 2, 2> level by level and k_jive2_coop<4>; the flat Jubjub / BN-254 batches they are compared with),
# on the GPU box, into gpurun_out/prof_cfg_<tag>/ -> profiles/<tag>/pmc_configs.json:
#   stats/                          rocprofv3 --kernel-trace --stats
#   pmc_fetch/ pmc_write/ pmc_sq/   three separate --pmc passes (never combined with trace domains)
#   tools/collect_config_profiles.sh <tag>
set -o pipefail
tag="${1:-r04}"
out="gpurun_out/prof_cfg_${tag}"
mkdir -p "$out"
export TMPDIR=/tmp
RUN="python3 tools/profile_workloads.py cfg3 cfg5 flat --reps 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- $RUN > "$out/stats.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -o run -- $RUN > "$out/pmc_fetch.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -o run -- $RUN > "$out/pmc_write.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv \
    -d "$out/pmc_sq" -o run -- $RUN > "$out/pmc_sq.log" 2>&1 || exit 1
python tools/summarize_config_profiles.py "$out" || exit 1
mkdir -p "profiles/${tag}" && cp "$out/pmc_configs.json" "$out/rocprofv3_kernel_stats_configs.csv" "profiles/${tag}/"
