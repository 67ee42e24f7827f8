#!/usr/bin/env python3
"""A small batch of LONG messages from host memory (>= 64 MiB in all: fed segment by segment, capi.hip sponge_segments):
which kernel absorbs the segments?  Round 4: always the lane-private ones (a few thousand messages = one wavefront per
16 SIMDs); round 5: the cooperative latency kernels, which now carry the sponge state between segments.
    ANEMOI_MI355X_LIB=<library> python tools/bench_segmented_latency.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
import numpy as np
import anemoi_amd as A

rng = np.random.default_rng(5)
print("library:", os.path.basename(A.lib_path()), "(laboratory build)" if A.is_ab_build() else "(product)")
for field, width, n, kib in (("bn_254", 4, 1024, 128), ("bn_254", 4, 4096, 32), ("jubjub", 2, 1024, 128), ("bls12_381", 2, 2048, 64)):
    msgs = rng.integers(0, 256, size=(n, kib * 1024), dtype=np.uint8)
    inst = A.Anemoi(field, width)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        d = inst.hash_batch(msgs)
        ts.append(time.perf_counter() - t0)
    print("%-10s %d-%d  %5d messages x %4d KiB (%4d MiB): %8.1f ms   digest xor %016x" % (
        field, width, width - 1, n, kib, n * kib // 1024, 1e3 * min(ts[1:]), int(np.bitwise_xor.reduce(d.reshape(-1)))))
