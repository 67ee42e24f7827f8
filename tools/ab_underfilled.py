#!/usr/bin/env python3
"""A/B of several builds on the launches that do NOT fill the machine, in ONE process on ONE device: config 3 (BN-254
4-3 sponge, 2^16 x 10 240 B = two wavefronts per SIMD), Jive batches of 2^15 .. 2^18 items (Jubjub, BLS12-381, BN-254
4-3), next to a full batch as the control.  Outputs of all builds must agree bit for bit.
    python tools/ab_underfilled.py name=path.so name=path.so ...
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
    lib.anemoi_merkle_root_dev.argtypes = [ci, vp, ctypes.c_uint, vp, vp, vp]
    libs.append((name, lib))
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream()
rng = np.random.default_rng(7)


def timed(fn, reps):
    ts = []
    for _ in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        assert fn() == 0
        b.record(s)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts[1:])


def ab(label, call, out, reps=3):
    res, ref = {}, None
    for rnd in range(2):
        for name, lib in libs:
            t = timed(lambda: call(lib), reps)
            res[name] = min(res.get(name, 1e30), t)
            got = out.clone()
            if ref is None:
                ref = got
            assert torch.equal(ref, got), (label, name)
    base = res[libs[0][0]]
    print("%-44s " % label + "   ".join("%s %9.3f ms (%+.1f %%)" % (n, res[n], (base / res[n] - 1) * 100) for n, _ in libs))


for field, name, limbs in ((4, "jubjub", 4), (0, "bls12_381", 6)):
    h = rng.integers(0, 1 << 60, size=(1 << 20, 2, limbs), dtype=np.uint64)
    d_in = torch.from_numpy(h.view(np.int64).reshape(-1)).to(dev)
    for lg in (15, 16, 17, 18, 20):
        n = 1 << lg
        d_out = torch.zeros(n * limbs, dtype=torch.int64, device=dev)
        ab("%s 2-1 Jive, 2^%d items" % (name, lg),
           lambda lib: lib.anemoi_jive_compress_k_dev(field, 2, 2, d_in.data_ptr(), d_out.data_ptr(), n, s.cuda_stream), d_out)
h = rng.integers(0, 1 << 60, size=(1 << 20, 4, 4), dtype=np.uint64)
d_in = torch.from_numpy(h.view(np.int64).reshape(-1)).to(dev)
for lg in (15, 16, 17, 20):
    n = 1 << lg
    d_out = torch.zeros(n * 8, dtype=torch.int64, device=dev)
    ab("bn_254 4-3 Jive, 2^%d items" % lg,
       lambda lib: lib.anemoi_jive_compress_k_dev(2, 4, 2, d_in.data_ptr(), d_out.data_ptr(), n, s.cuda_stream), d_out)
nmsg = 1 << 16
msgs = torch.from_numpy(rng.integers(0, 256, size=(nmsg, 10240), dtype=np.uint8)).to(dev)
dig = torch.zeros(nmsg * 4, dtype=torch.int64, device=dev)
ab("config 3: bn_254 4-3 sponge, 2^16 x 10240 B",
   lambda lib: lib.anemoi_hash_bytes_dev(2, 4, msgs.data_ptr(), 10240, nmsg, dig.data_ptr(), s.cuda_stream), dig, reps=2)
depth = 21
h = rng.integers(0, 1 << 60, size=(1 << depth, 1, 4), dtype=np.uint64)
leaves = torch.from_numpy(h.view(np.int64).reshape(-1)).to(dev)
scratch = torch.empty((1 << depth) * 4, dtype=torch.int64, device=dev)
root = torch.zeros(4, dtype=torch.int64, device=dev)
ab("config 5: jubjub depth-21 subtree",
   lambda lib: lib.anemoi_merkle_root_dev(4, leaves.data_ptr(), depth, scratch.data_ptr(), root.data_ptr(), s.cuda_stream), root, reps=2)
