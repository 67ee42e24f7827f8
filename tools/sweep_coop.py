#!/usr/bin/env python3
"""Which Jive 2-1 kernel for which batch size?  Times the four kernels -- two-row fold (two items per wavefront, coop2d.h),
row-cooperative scan (four items per wavefront, one per 16-lane DPP row), the one-item-per-wavefront scan (A/B) and
lane-private (one item per lane) -- on device-resident batches of 1 .. 65 536 items in ONE process (kernels are chosen
through anemoi_set_option), checks that all agree bit for bit, and prints the per-size table plus the Merkle-tree
time each cut-off policy would give (a depth-d tree is one launch per level: 2^(d-1), .., 2, 1 items).

    python tools/sweep_coop.py [field ...]            (default: jubjub bls12_381)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
import numpy as np
import torch
import anemoi_amd as A
from anemoi_amd import synth

fields = [a for a in sys.argv[1:]] or ["jubjub", "bls12_381"]
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream()
BIG = 1000000000
# mode -> (coop_max, coop2d_max, coop4_max)
MODES = {"coop2d": (0, BIG, 0), "coop4": (0, 0, BIG), "coop1": (BIG, 0, 0), "lane": (0, 0, 0)}
AB = A.is_ab_build()   # "coop1" (one item per wavefront) exists in `make AB=1` libraries only: ANEMOI_MI355X_LIB=.../libanemoi_ab.so


def timed(fid, d_in, d_out, n, reps=5):
    ts = []
    for _ in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        assert A.lib.anemoi_jive_compress_k_dev(fid, 2, 2, d_in.data_ptr(), d_out.data_ptr(), n, s.cuda_stream) == 0
        b.record(s)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts[1:])[len(ts[1:]) // 2]


for field in fields:
    fid, L = A.field_id(field), synth.limbs_of(field)
    sizes = [1 << k for k in range(0, 17)]
    st = synth.states(field, 2, 0xC0, 0, sizes[-1])
    d_in = torch.from_numpy(st.view(np.int64).reshape(-1)).to(dev)
    table = {}
    print("%s: kernel time in ms per batch size (median of 5, device-resident)" % field)
    print("%8s %9s %9s %9s %9s   best" % ("items", "coop2d", "coop4", "coop1", "lane"))
    for n in sizes:
        row, ref = {}, None
        for mode, (c1, c2, c4) in MODES.items():
            if (mode == "coop1" and (n > 4096 or not AB)) or (mode == "coop2d" and n > 16384):
                continue     # tens of thousands of wavefronts of a latency kernel are pointless
            if AB:
                A.set_option("coop_max", c1)
            A.set_option("coop2d_max", c2); A.set_option("coop4_max", c4)
            d_out = torch.zeros(n * L, dtype=torch.int64, device=dev)
            row[mode] = timed(fid, d_in, d_out, n)
            got = d_out.cpu()
            if ref is None:
                ref = got
            assert torch.equal(ref, got), (field, n, mode)
        table[n] = row
        best = min(row, key=row.get)
        print("%8d %9s %9.3f %9s %9.3f   %s" % (n, "%9.3f" % row["coop2d"] if "coop2d" in row else "-", row["coop4"],
                                               "%9.3f" % row["coop1"] if "coop1" in row else "-", row["lane"], best))
    for o in (("coop_max",) if AB else ()) + ("coop2d_max", "coop4_max"):
        A.set_option(o, None)

    def tree_ms(depth, c2max, c4max):
        t = 0.0
        for l in range(depth):
            n = 1 << (depth - 1 - l)
            if n > sizes[-1]:
                continue     # levels above the sweep are lane-private in every policy
            mode = "coop2d" if n <= c2max else ("coop4" if n <= c4max else "lane")
            t += table[n][mode]
        return t

    depth = 17   # levels of 65 536 .. 1 items: the part of any deeper tree the choice affects
    print("  levels of <= 65 536 items of a tree (17 launches), by cut-off policy (coop2d up to / coop4 up to):")
    for c2max in (0, 1024, 2048, 4096, 8192):
        for c4max in (0, 4096, 8192, 16384):
            if c4max and c4max <= c2max:
                continue
            print("    coop2d <= %5d, coop4 <= %5d: %7.2f ms" % (c2max, c4max, tree_ms(depth, c2max, c4max)))

# ---- Anemoi-4-3: two-row fold (one state per wavefront) / row-cooperative scan (two) / the lane-pair kernel ---------------------------
for field in [f for f in fields if f in ("bn_254", "bls12_381", "jubjub")] or ["bn_254", "bls12_381"]:
    fid, L = A.field_id(field), synth.limbs_of(field)
    sizes = [1 << k for k in range(0, 15)]
    st = synth.states(field, 4, 0xC4, 0, sizes[-1])
    d_in = torch.from_numpy(st.view(np.int64).reshape(-1)).to(dev)
    print("%s Anemoi-4-3 Jive (k = 2): kernel time in ms per batch size" % field)
    print("%8s %10s %10s %10s   best" % ("states", "two-row", "row-coop", "lane-pair"))

    def timed43(n, d_out):
        ts = []
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            assert A.lib.anemoi_jive_compress_k_dev(fid, 4, 2, d_in.data_ptr(), d_out.data_ptr(), n, s.cuda_stream) == 0
            b.record(s)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return sorted(ts[1:])[2]

    for n in sizes:
        row, ref = {}, None
        for mode, cmax, c2d in (("two-row", 0, BIG), ("row-coop", BIG, 0), ("lane-pair", 0, 0)):
            if mode == "two-row" and n > 8192:
                continue
            A.set_option("coop43_max", cmax)
            A.set_option("coop2d43_max", c2d)
            d_out = torch.zeros(n * 2 * L, dtype=torch.int64, device=dev)
            row[mode] = timed43(n, d_out)
            got = d_out.cpu()
            if ref is None:
                ref = got
            assert torch.equal(ref, got), (field, n, mode)
        print("%8d %10s %10.3f %10.3f   %s" % (n, "%.3f" % row["two-row"] if "two-row" in row else "-", row["row-coop"],
                                              row["lane-pair"], min(row, key=row.get)))
    A.set_option("coop43_max", None)
    A.set_option("coop2d43_max", None)

# ---- sponge: row-cooperative (4 / 2 messages per wavefront) against the lane-private kernels, 1 KB messages ----------
for field, width in (("jubjub", 2), ("bn_254", 4), ("bls12_381", 2)):
    if field not in fields and fields != ["jubjub", "bls12_381"]:
        continue
    fid, L = A.field_id(field), synth.limbs_of(field)
    lib = A.lib
    import ctypes
    lib.anemoi_hash_bytes_dev.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                          ctypes.c_void_p, ctypes.c_void_p]
    mlen = 1024
    sizes = [1 << k for k in range(0, 15, 2)]
    msgs = torch.from_numpy(np.random.default_rng(5).integers(0, 256, size=(sizes[-1], mlen), dtype=np.uint8)).to(dev)
    print("%s Anemoi-%s sponge, %d-byte messages: kernel time in ms per batch size" % (field, "2-1" if width == 2 else "4-3", mlen))
    print("%8s %10s %10s %10s   best" % ("messages", "two-row", "row-coop", "lane"))
    for n in sizes:
        row, ref = {}, None
        for mode, cmax, c2 in (("two-row", BIG, BIG), ("row-coop", BIG, 0), ("lane", 0, 0)):
            if mode == "two-row" and width == 4 and n > 4096:
                continue
            A.set_option("coop_sponge_max", cmax); A.set_option("coop2d_max", c2); A.set_option("coop2d43_max", c2)
            d_out = torch.zeros(n * L, dtype=torch.int64, device=dev)
            ts = []
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(s)
                assert lib.anemoi_hash_bytes_dev(fid, width, msgs.data_ptr(), mlen, n, d_out.data_ptr(), s.cuda_stream) == 0
                b.record(s)
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            row[mode] = min(ts[1:])
            got = d_out.cpu()
            if ref is None:
                ref = got
            assert torch.equal(ref, got), (field, n, mode)
        print("%8d %10s %10.3f %10.3f   %s" % (n, "%10.3f" % row["two-row"] if "two-row" in row else "-", row["row-coop"], row["lane"], min(row, key=row.get)))
    A.set_option("coop_sponge_max", None); A.set_option("coop2d_max", None); A.set_option("coop2d43_max", None)
