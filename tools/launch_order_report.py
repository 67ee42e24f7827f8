#!/usr/bin/env python3
"""Per-DISPATCH reading of one rocprofv3 --pmc pass (GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
[SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU]) in launch order: what distinguishes a slow dispatch of a kernel
from a fast dispatch of the SAME kernel on the same grid --

  clock GHz        = GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md, DVFS give-back: rocprofv3 reports the
                     sum over the XCDs; good to ~3 % on dispatches of >= 10 ms)
  Mcycles          = GRBM_GUI_ACTIVE / 8: a dispatch that is slow at EQUAL cycles was slow because of the clock
  waves / SIMD     = SQ_WAVE_CYCLES x 4 / (cycles x 1 024 SIMDs): average residency
  busy CUs         = SQ_BUSY_CU_CYCLES / (cycles x 256 CUs) where collected: < 1 means part of the chip held no wavefront
  cycles / VALU    = cycles x 1 024 / (SQ_INSTS_VALU): per SIMD

    python tools/launch_order_report.py <dir with *counter_collection.csv> [--min-ms 1.0]
"""
import collections
import csv
import glob
import os
import re
import sys

XCDS, SIMDS, CUS = 8, 1024, 256


def main():
    d = sys.argv[1]
    min_ms = float(sys.argv[sys.argv.index("--min-ms") + 1]) if "--min-ms" in sys.argv else 1.0
    hits = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
    if not hits:
        sys.exit("no counter_collection.csv under " + d)
    per = collections.OrderedDict()
    with open(hits[0], newline="") as f:
        for r in csv.DictReader(f):
            key = int(r["Dispatch_Id"])
            m = re.search(r"(k_[a-z0-9_]+<[^>]*>)", r["Kernel_Name"])
            name = m.group(1) if m else r["Kernel_Name"].split("(")[0][-40:]
            e = per.setdefault(key, {"kernel": name, "grid": int(r["Grid_Size"]), "start": int(r["Start_Timestamp"]),
                                     "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    t0 = min(e["start"] for e in per.values())
    print("%4s %-28s %9s %10s %9s %7s %9s %9s %8s %9s" % ("disp", "kernel", "grid", "start ms", "ms", "GHz", "Mcycles",
                                                          "waves/SIMD", "busy CUs", "cyc/VALU"))
    for k, e in sorted(per.items()):
        if e["ms"] < min_ms or "GRBM_GUI_ACTIVE" not in e:
            continue
        cyc = e["GRBM_GUI_ACTIVE"] / XCDS
        wps = e.get("SQ_WAVE_CYCLES", 0.0) * 4 / (cyc * SIMDS)
        bcu = ("%8.3f" % (e["SQ_BUSY_CU_CYCLES"] / (cyc * CUS))) if "SQ_BUSY_CU_CYCLES" in e else "       -"
        cpv = ("%9.2f" % (cyc * SIMDS / e["SQ_INSTS_VALU"])) if e.get("SQ_INSTS_VALU") else "        -"
        print("%4d %-28s %9d %10.1f %9.2f %7.3f %9.2f %9.2f %s %s" % (k, e["kernel"][:28], e["grid"], (e["start"] - t0) / 1e6,
                                                                     e["ms"], cyc / e["ms"] / 1e6, cyc / 1e6, wps, bcu, cpv))


if __name__ == "__main__":
    main()
