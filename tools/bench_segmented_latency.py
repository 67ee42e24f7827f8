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
import ctypes
import numpy as np

path = os.environ.get("ANEMOI_MI355X_LIB", os.path.join(ROOT, "anemoi-rust_amd", "lib", "libanemoi_mi355x.so"))
lib = ctypes.CDLL(path)      # raw ctypes: older libraries (round 4's) lack symbols the Python package binds
fn = lib.anemoi_hash_bytes_batch
fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
rng = np.random.default_rng(5)
print("library:", os.path.basename(path))
FIELDS = ["bls12_381", "bls12_377", "bn_254", "ed_on_bls12_377", "jubjub", "pallas", "vesta"]
for field, width, n, kib in (("bn_254", 4, 1024, 128), ("bn_254", 4, 4096, 32), ("jubjub", 2, 1024, 128), ("bls12_381", 2, 2048, 64)):
    fid, limbs = FIELDS.index(field), 6 if field.startswith("bls12_38") or field == "bls12_377" else 4
    msgs = rng.integers(0, 256, size=(n, kib * 1024), dtype=np.uint8)
    d = np.zeros((n, limbs), dtype=np.uint64)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        assert fn(fid, width, msgs.ctypes.data, kib * 1024, n, d.ctypes.data, 0) == 0
        ts.append(time.perf_counter() - t0)
    print("%-10s %d-%d  %5d messages x %4d KiB (%4d MiB): %8.1f ms   digest xor %016x" % (
        field, width, width - 1, n, kib, n * kib // 1024, 1e3 * min(ts[1:]), int(np.bitwise_xor.reduce(d.reshape(-1)))))
