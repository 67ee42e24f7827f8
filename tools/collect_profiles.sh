#!/bin/bash
# Collects the judged evidence for the headline kernel on the GPU box, into gpurun_out/prof_<tag>/:
#   bench_n1.json                 python bench.py (default flags)
#   stats/                        rocprofv3 --kernel-trace --stats of the same bench command
#   pmc_fetch/ pmc_write/ pmc_sq/ three separate --pmc passes (never combined with trace domains)
# then tools/summarize_profiles.py turns them into the files kept under profiles/rNN/.
#   tools/collect_profiles.sh <tag>
set -o pipefail
tag="${1:-r04}"
out="gpurun_out/prof_${tag}"
mkdir -p "$out"
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline"
# a first bench line for the summariser (batch size, kernel time); the judged line is taken at the end, after the
# fresh counter summary is in place, so that its roofline.traffic / alu.valu_issue_frac refer to THIS build
python bench.py --no-cpu-baseline --steps 2 > "$out/bench_n1.json" 2> "$out/bench_n1.err" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- $BENCH > "$out/stats.log" 2>&1 || exit 1
# (without the clock sampler: under --pmc kernels run one at a time, and the counters of a dispatch are the whole chip's)
PMC="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-clock-sampler"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -o run -- $PMC > "$out/pmc_fetch.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -o run -- $PMC > "$out/pmc_write.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv \
    -d "$out/pmc_sq" -o run -- $PMC > "$out/pmc_sq.log" 2>&1 || exit 1
python tools/summarize_profiles.py "$out" || exit 1
mkdir -p "profiles/${tag}" && cp "$out/pmc_k_jive.json" "profiles/${tag}/pmc_k_jive.json"
python bench.py > "$out/bench_n1.json" 2> "$out/bench_n1.err" || exit 1
tail -c 600 "$out/bench_n1.json"; echo
