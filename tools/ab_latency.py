#!/usr/bin/env python3
"""A/B of several builds on the LATENCY side, in one process: one / 1 024 / 4 096 Jive 2-1 compressions (the
row-cooperative kernel), 1 024 4-3 states, 64 messages of 1 KB through the cooperative sponge.
    python tools/ab_latency.py name=path.so ...
"""
import ctypes
import os
import sys

import numpy as np
import torch

vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
libs = []
for spec in sys.argv[1:]:
    name, path = spec.split("=", 1)
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.anemoi_jive_compress_k_dev.argtypes = [ci, ci, ci, vp, vp, sz, vp]
    lib.anemoi_hash_bytes_dev.argtypes = [ci, ci, vp, sz, sz, vp, vp]
    libs.append((name, lib))
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream()
rng = np.random.default_rng(7)


def timed(fn, reps=7):
    ts = []
    for _ in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        assert fn() == 0
        b.record(s)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts[1:])[len(ts[1:]) // 2]


def ab(label, call, out):
    res, ref = {}, None
    for rnd in range(2):
        for name, lib in libs:
            t = timed(lambda: call(lib))
            res[name] = min(res.get(name, 1e30), t)
            got = out.clone()
            if ref is None:
                ref = got
            assert torch.equal(ref, got), (label, name)
    base = res[libs[0][0]]
    print("%-44s " % label + "   ".join("%s %8.3f ms (%+.1f %%)" % (n, res[n], (base / res[n] - 1) * 100) for n, _ in libs))


for field, name, limbs in ((4, "jubjub", 4), (0, "bls12_381", 6), (6, "vesta", 4)):
    h = rng.integers(0, 1 << 60, size=(4096, 2, limbs), dtype=np.uint64)
    d_in = torch.from_numpy(h.view(np.int64).reshape(-1)).to(dev)
    for n in (1, 1024, 4096):
        d_out = torch.zeros(n * limbs, dtype=torch.int64, device=dev)
        ab("%s 2-1 Jive, %d items" % (name, n),
           lambda lib: lib.anemoi_jive_compress_k_dev(field, 2, 2, d_in.data_ptr(), d_out.data_ptr(), n, s.cuda_stream), d_out)
h = rng.integers(0, 1 << 60, size=(1024, 4, 4), dtype=np.uint64)
d_in = torch.from_numpy(h.view(np.int64).reshape(-1)).to(dev)
d_out = torch.zeros(1024 * 8, dtype=torch.int64, device=dev)
ab("bn_254 4-3 Jive, 1024 states", lambda lib: lib.anemoi_jive_compress_k_dev(2, 4, 2, d_in.data_ptr(), d_out.data_ptr(), 1024, s.cuda_stream), d_out)
msgs = torch.from_numpy(rng.integers(0, 256, size=(64, 1024), dtype=np.uint8)).to(dev)
dig = torch.zeros(64 * 4, dtype=torch.int64, device=dev)
ab("jubjub 2-1 sponge, 64 x 1 KB", lambda lib: lib.anemoi_hash_bytes_dev(4, 2, msgs.data_ptr(), 1024, 64, dig.data_ptr(), s.cuda_stream), dig)
