#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of tools/collect_profiles.sh into the summaries kept under profiles/:
  rocprofv3_kernel_stats_bench.csv   the --stats table
  rocprofv3_kernel_trace_k_jive.csv  the headline kernel's dispatch rows (VGPR / LDS / duration)
  pmc_k_jive.json                    per-launch counter averages + derived figures (HBM traffic with the
                                     gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md, VALU busy
                                     fraction, multiply-add issue rate against the measured peak)
    python tools/summarize_profiles.py gpurun_out/prof_<tag>
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd", "anemoi_amd"))
import buildinfo  # noqa: E402  (plain module import: the package itself needs the HIP library)

KERNEL = "k_jive"
CUS, SIMDS = 256, 4
XCDS = 8                 # GRBM_GUI_ACTIVE is reported once per XCD; the sum is divided by this
# cycles per wave-instruction per SIMD used by the issue model: 16 lanes per clock for the multiply-add,
# the measured costs of tools/ubench/wall_rates.hip for the other classes (profiles/r01/ubench_wall_rates.txt)
C_MAD, C_WIDE, C_SIMPLE = 4.0, 4.3, 2.4


def find(d, pat):
    hits = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    return hits[0] if hits else None


def rows(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def counters(d):
    """per dispatch of the headline kernel, summed over a counter's hardware instances: the MEDIAN over the dispatches of
    the pass (rocprofv3 serialises the kernels of ONE queue; the counters of the last k_jive dispatch can take in the first
    waves of a kernel that another queue starts right behind it -- bench.py's probes run on the library's own stream --
    e.g. SQ_WAVES 19 456 = 16 384 + 3 072 in one dispatch of round 5's collection)"""
    path = find(d, "*counter_collection.csv")
    per = {}
    for r in rows(path):
        if KERNEL not in r["Kernel_Name"] or "coop" in r["Kernel_Name"]:
            continue
        key = (r["Counter_Name"], r["Dispatch_Id"])
        per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
    out = {}
    for (name, _), v in per.items():
        out.setdefault(name, []).append(v)
    return {k: sorted(v)[len(v) // 2] for k, v in out.items()}


def mad_counts():
    """v_mad_u64_u32 per squaring / multiplication of the shipped BLS12-381 layout, from the generated header"""
    return buildinfo.bls12_381_limb_layout()


def main():
    d = sys.argv[1]
    dst = d
    stats = find(os.path.join(d, "stats"), "*kernel_stats.csv")
    trace = find(os.path.join(d, "stats"), "*kernel_trace.csv")
    with open(os.path.join(dst, "rocprofv3_kernel_stats_bench.csv"), "w") as f:
        f.write(open(stats).read())
    tr = [r for r in rows(trace) if KERNEL in r["Kernel_Name"] and "coop" not in r["Kernel_Name"]]
    with open(os.path.join(dst, "rocprofv3_kernel_trace_k_jive.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(tr[0].keys()))
        w.writeheader()
        w.writerows(tr)
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in tr]
    c = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        c.update(counters(os.path.join(d, sub)))
    sq_trace = find(os.path.join(d, "pmc_sq"), "*kernel_trace.csv")
    secs = None
    if sq_trace:
        t = [r for r in rows(sq_trace) if KERNEL in r["Kernel_Name"] and "coop" not in r["Kernel_Name"]]
        if t:
            secs = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in t) / len(t) / 1e9
    bench = json.loads(open(os.path.join(d, "bench_n1.json")).read().strip().splitlines()[-1])
    if secs is None:
        secs = bench["roofline"]["kernel_ms"] / 1e3
    n = bench["config"]["batch_per_gpu"]
    fetch = c["FETCH_SIZE"] * 1024 * 2   # KiB; gfx950 reports half of a wide coalesced read
    write = c["WRITE_SIZE"] * 1024
    mc = mad_counts()
    rounds = 21
    sq_per_round, mul_per_round = buildinfo.bls12_381_products_per_round()   # S-box chain + its 2 squarings; + 2 settle()
    mad_per_comp = buildinfo.bls12_381_mad_per_compression()
    wide_per_comp = rounds * (sq_per_round * mc["sqr_wide"] + mul_per_round * mc["mul_wide"]) + 5 * mc["mul_wide"]
    gui = c["GRBM_GUI_ACTIVE"] / XCDS
    clock = gui / secs / 1e9
    valu_per_wave = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
    simd_cycles = gui * CUS * SIMDS                  # SIMD-cycles available during one launch
    waves_per_simd = c["SQ_WAVES"] / (CUS * SIMDS)   # wavefronts each SIMD executes per launch
    simple_per_wave = valu_per_wave - mad_per_comp - wide_per_comp
    derived = {
        "hbm_traffic_bytes_per_launch": fetch + write,
        "fetch_bytes_corrected_x2": fetch,
        "write_bytes": write,
        "algorithmic_bytes_per_launch": 144 * n,
        "clock_GHz": clock,
        "avg_waves_per_SIMD": c["SQ_WAVE_CYCLES"] * 4 / simd_cycles,
        "valu_wave_instructions_per_wavefront": valu_per_wave,
        "valu_wave_instructions_per_compression": valu_per_wave / 64,
        "cycles_per_valu_instr_per_SIMD": simd_cycles / c["SQ_INSTS_VALU"],
        "mad_wave_instructions_per_wavefront": mad_per_comp,
        "mad_fraction_of_valu_instructions": mad_per_comp / valu_per_wave,
        # share of all SIMD cycles spent issuing v_mad_u64_u32 at 16 lanes per clock (4 cycles per wave-instruction)
        "mad_cycle_fraction": mad_per_comp * C_MAD * waves_per_simd / gui,
        # all VALU work priced per class: multiply-adds, 64-bit shift/add + v_mul_lo_u32, everything else
        "valu_issue_model_fraction": (mad_per_comp * C_MAD + wide_per_comp * C_WIDE + simple_per_wave * C_SIMPLE)
        * waves_per_simd / gui,
        # raw counter ratio; exceeds 1 because simple VALU ops issue in fewer than 4 cycles on gfx950
        "valu_quadcycles_over_simd_cycles": c["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles,
    }
    c["kernel_seconds_in_sq_pass"] = secs
    out = {
        "csrc_sha256": buildinfo.csrc_sha256(),   # bench.py reports `traffic` only for the sources this describes
        "products_per_round": {"squarings": sq_per_round, "multiplications": mul_per_round},
        # the dispatch record of the headline kernel as rocprofv3 reports it (its VGPR_Count is in allocation units of 2
        # registers on gfx950 and LDS_Block_Size does not include dynamic LDS: the kernel has 167 VGPRs, 12 KiB of LDS)
        "kernel_dispatch": {k: tr[0][k] for k in ("Kernel_Name", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                                    "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size_X", "Grid_Size_X")
                            if k in tr[0]},
        "counters_avg_per_launch": c, "derived": derived, "limb_layout": mc,
        "rocprofv3_stats_avg_ms": sum(dur) / len(dur), "rocprofv3_stats_calls": len(dur),
        "note": "rocprofv3 --pmc passes (separate runs for FETCH_SIZE, WRITE_SIZE and the SQ/GRBM set) of `python3 "
                "bench.py --steps 2 --warmup 1 --no-cpu-baseline` (tools/collect_profiles.sh); values are medians over "
                "the k_jive dispatches, summed over a counter's hardware instances. FETCH_SIZE/WRITE_SIZE are in KiB; "
                "FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read). "
                "GRBM_GUI_ACTIVE is summed over the 8 XCDs. SQ_WAVE_CYCLES / SQ_ACTIVE_INST_VALU are quad-cycles. mad_* "
                "figures: v_mad_u64_u32 counts from the generated assembly (limb_layout), 21 x (products_per_round) + 5 "
                "per compression (anemoi_amd/buildinfo.py). Issue model: 4.0 cycles per wave-level v_mad_u64_u32 (16 lanes per "
                "clock), 4.3 for 64-bit shift/add and v_mul_lo_u32, 2.4 for the remaining VALU instructions "
                "(tools/ubench/wall_rates.hip).",
    }
    with open(os.path.join(dst, "pmc_k_jive.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: derived[k] for k in ("clock_GHz", "cycles_per_valu_instr_per_SIMD", "mad_fraction_of_valu_instructions",
                                              "mad_cycle_fraction", "valu_issue_model_fraction",
                                              "valu_quadcycles_over_simd_cycles", "hbm_traffic_bytes_per_launch")}, indent=1))
    print("rocprofv3 --stats: %d calls, avg %.2f ms" % (len(dur), sum(dur) / len(dur)))


if __name__ == "__main__":
    main()
