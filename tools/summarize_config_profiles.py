#!/usr/bin/env python3
"""Summarises the raw rocprofv3 output of tools/collect_config_profiles.sh into pmc_configs.json: one record per
(kernel, grid size) of the non-headline workloads, with what the review asked for --

  * VALU instructions per item (SQ_INSTS_VALU / SQ_WAVES / items per wavefront),
  * resident waves per SIMD averaged over the launch (SQ_WAVE_CYCLES * 4 / SIMD-cycles),
  * HBM traffic (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes, the gfx950 correction of MI355X_MICROARCH.md) next to
    the algorithmic bytes of the launch (config 3: 10 272 B per message; a merge: 96 B),
  * cycles per VALU instruction per SIMD, kernel time, and the rate as a fraction of the FLAT rate of the same
    instance in the same session (config 3 vs 2^20 BN-254 4-3 compressions, per permutation; config 5's levels vs
    2^20 Jubjub 2-1 compressions).

    python tools/summarize_config_profiles.py gpurun_out/prof_cfg_<tag>
"""
import csv
import glob
import json
import os
import re
import sys

CUS, SIMDS, XCDS = 256, 4, 8


def find(d, pat):
    hits = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    return hits[0] if hits else None


def rows(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def short(name):
    m = re.search(r"anemoi::(k_[a-z0-9_]+<[^>]*>)", name)
    return m.group(1) if m else None


def counters(d):
    """{(kernel, grid): {counter: average per dispatch}} and the dispatch durations seen in that pass"""
    per, meta = {}, {}
    for r in rows(find(d, "*counter_collection.csv")):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        key = (k, int(r["Grid_Size"]), r["Dispatch_Id"])
        per.setdefault(key, {})
        per[key][r["Counter_Name"]] = per[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        meta[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    # per (kernel, grid) the FASTEST dispatch of the pass: the first dispatch of a workload pays first-touch effects
    # (config 3's first launch over its fresh 640 MiB input: 511 ms against 358 ms)
    out = {}
    for (k, g, d), c in per.items():
        ms = meta[(k, g, d)]
        if (k, g) not in out or ms < out[(k, g)]["ms"]:
            out[(k, g)] = dict(c, ms=ms)
    for (k, g), o in out.items():
        o["dispatches"] = sum(1 for (k2, g2, _) in per if (k2, g2) == (k, g))
    return out


def items_per_wave(kernel):
    if "coop" in kernel:   # row-cooperative scan: four items per wavefront; two-row fold: two; one item per wavefront else
        return 4 if kernel.endswith(", 16>") else (2 if kernel.endswith(", 32>") else 1)
    return 32 if ("_pair" in kernel) else 64


def algorithmic_bytes(kernel, items):
    if kernel.startswith("k_sponge_pair<2"):
        return 10272 * items           # config 3: 10 240 message bytes + a 32-byte digest
    if kernel.startswith("k_jive<4") or kernel.startswith("k_jive2_coop<4"):
        return 96 * items              # Jubjub merge: 64 B in + 32 B out (the cooperative kernel's partly filled last wavefront counted whole)
    if kernel.startswith("k_jive_pair<2"):
        return 192 * items             # BN-254 4-3 Jive: 128 B in + 64 B out
    return None


def main():
    d = sys.argv[1]
    stats = find(os.path.join(d, "stats"), "*kernel_stats.csv")
    with open(os.path.join(d, "rocprofv3_kernel_stats_configs.csv"), "w") as f:
        f.write(open(stats).read())
    trace = rows(find(os.path.join(d, "stats"), "*kernel_trace.csv"))
    dur = {}
    for r in trace:
        k = short(r["Kernel_Name"])
        if k:
            dur.setdefault((k, int(r["Grid_Size_X"])), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    c = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        for key, v in counters(os.path.join(d, sub)).items():
            c.setdefault(key, {}).update({k: x for k, x in v.items() if k not in ("ms", "dispatches")})
            if sub == "pmc_sq":
                c[key]["ms_in_sq_pass"] = v["ms"]
    recs = []
    for (k, grid), v in sorted(c.items()):
        waves = v.get("SQ_WAVES", grid / 64)
        ipw = items_per_wave(k)
        items = int(round(waves * ipw))
        ds = sorted(dur.get((k, grid), []))
        ms = ds[0] if ds else None      # the fastest dispatch: the first one of a workload pays first-touch effects
        rec = {"kernel": k, "grid": grid, "wavefronts": waves, "kernel_ms_min_stats_pass": ms, "kernel_ms_all": ds,
               "dispatches_timed": len(ds)}
        if "GRBM_GUI_ACTIVE" in v:
            gui = v["GRBM_GUI_ACTIVE"] / XCDS
            simd_cycles = gui * CUS * SIMDS
            rec.update({
                "clock_GHz": gui / (v["ms_in_sq_pass"] / 1e3) / 1e9,
                "valu_instr_per_wavefront": v["SQ_INSTS_VALU"] / waves,
                "valu_instr_per_item": v["SQ_INSTS_VALU"] / waves / ipw,
                "avg_waves_per_SIMD": v["SQ_WAVE_CYCLES"] * 4 / simd_cycles,
                "cycles_per_valu_instr_per_SIMD": simd_cycles / v["SQ_INSTS_VALU"],
                "salu_instr_per_wavefront": v["SQ_INSTS_SALU"] / waves,
                "lds_instr_per_wavefront": v["SQ_INSTS_LDS"] / waves,
            })
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            rec["hbm_traffic_bytes"] = 2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024
            if items:
                ab = algorithmic_bytes(k, items if "sponge" not in k else int(round(waves * ipw)))
                if ab:
                    rec["algorithmic_bytes"] = ab
                    rec["traffic_over_algorithmic"] = rec["hbm_traffic_bytes"] / ab
        recs.append(rec)

    def med(kernel, pred=lambda g: True):
        xs = [r for r in recs if r["kernel"] == kernel and pred(r["grid"]) and r["kernel_ms_min_stats_pass"]]
        return xs[0] if xs else None

    summary = {}
    flat_j, flat_b = med("k_jive<4, 2, 2>", lambda g: g == (1 << 20)), med("k_jive_pair<2, 2>")
    if flat_j:
        summary["flat_jubjub_2_1_M_per_s"] = (1 << 20) / flat_j["kernel_ms_min_stats_pass"] / 1e3
    if flat_b:
        summary["flat_bn254_4_3_M_per_s"] = (1 << 20) / flat_b["kernel_ms_min_stats_pass"] / 1e3
    sp = med("k_sponge_pair<2, true>")
    if sp and flat_b:
        perm_rate = (1 << 16) * 111 / sp["kernel_ms_min_stats_pass"] / 1e3      # M permutations / s
        summary["cfg3_ms"] = sp["kernel_ms_min_stats_pass"]
        summary["cfg3_fraction_of_flat_rate"] = perm_rate / summary["flat_bn254_4_3_M_per_s"]
    if flat_j:
        # the depth-21 tree: levels of 2^20 .. 2^14 nodes on k_jive (grid = nodes), 2^13 and 2^12 on the row-cooperative
        # scan kernel (four nodes per wavefront: grid = ceil(nodes / 4) * 64), 2^11 .. 1 on the two-row fold kernel (two
        # nodes per wavefront)
        tree_ms = 0.0
        levels = []
        for l in range(21):
            nodes = 1 << (20 - l)
            if nodes > 8192:
                r = med("k_jive<4, 2, 2>", lambda g, nodes=nodes: g == nodes)
            elif nodes > 2048:
                r = med("k_jive2_coop<4, 16>", lambda g, nodes=nodes: g == (nodes + 3) // 4 * 64)
            else:
                r = med("k_jive2_coop<4, 32>", lambda g, nodes=nodes: g == (nodes + 1) // 2 * 64)
            if r:
                levels.append({"nodes": nodes, "kernel": r["kernel"], "ms": r["kernel_ms_min_stats_pass"],
                               "avg_waves_per_SIMD": r.get("avg_waves_per_SIMD")})
                tree_ms += r["kernel_ms_min_stats_pass"]
        summary["cfg5_levels"] = levels
        summary["cfg5_sum_of_levels_ms"] = tree_ms
        if tree_ms:
            summary["cfg5_fraction_of_flat_rate"] = ((1 << 21) - 1) / tree_ms / 1e3 / summary["flat_jubjub_2_1_M_per_s"]
    out = {"summary": summary, "kernels": recs,
           "note": "rocprofv3 passes of `python3 tools/profile_workloads.py cfg3 cfg5 flat --reps 2` "
                   "(tools/collect_config_profiles.sh): --kernel-trace --stats for the times, separate --pmc passes for "
                   "FETCH_SIZE, WRITE_SIZE and the SQ/GRBM set; counters and times are those of the FASTEST dispatch of a (kernel, "
                   "grid) pair in each pass (the first dispatch of a workload pays first-touch effects), summed over a counter's hardware instances; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled "
                   "(gfx950 reports half of a wide coalesced read); GRBM_GUI_ACTIVE summed over the 8 XCDs; "
                   "SQ_WAVE_CYCLES in quad-cycles."}
    with open(os.path.join(d, "pmc_configs.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(summary, indent=1)[:3000])


if __name__ == "__main__":
    main()
