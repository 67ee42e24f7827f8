#!/usr/bin/env python3
"""A/B of the balanced-placement LDS request (anemoi_kernels.h balanced_lds) in ONE process: the knob
ANEMOI_BALANCE_LDS is read at every launch.  Launches that do not fill the machine -- config 3 (2 048 wavefronts on
1 024 SIMDs), Jive batches of 2^13 .. 2^19 items -- with and without it.
    python tools/ab_balance.py [--lib path.so]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=os.path.join(ROOT, "anemoi-rust_amd", "lib", "libanemoi_mi355x.so"))
args = ap.parse_args()
lib = ctypes.CDLL(os.path.abspath(args.lib))
vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
lib.anemoi_jive_compress_k_dev.argtypes = [ci, ci, ci, vp, vp, sz, vp]
lib.anemoi_hash_bytes_dev.argtypes = [ci, ci, vp, sz, sz, vp, vp]
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream()
rng = np.random.default_rng(7)


def timed(fn, reps=5):
    ts = []
    for _ in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        assert fn() == 0
        b.record(s)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts[1:])[len(ts[1:]) // 2]


def ab(label, fn, out):
    res = {}
    ref = None
    for mode in ("0", "1", "0", "1"):
        os.environ["ANEMOI_BALANCE_LDS"] = mode
        res.setdefault(mode, []).append(timed(fn))
        got = out.clone()
        if ref is None:
            ref = got
        assert torch.equal(ref, got)
    off, on = min(res["0"]), min(res["1"])
    print("%-44s plain %8.3f ms   balanced %8.3f ms   (%+.1f %%)" % (label, off, on, (off / on - 1) * 100))


for field, name, limbs in ((4, "jubjub", 4), (0, "bls12_381", 6)):
    n_max = 1 << 19
    h = rng.integers(0, 1 << 60, size=(n_max, 2, limbs), dtype=np.uint64)
    d_in = torch.from_numpy(h.view(np.int64).reshape(-1)).to(dev)
    for lg in (14, 15, 16, 17, 18, 19):
        n = 1 << lg
        d_out = torch.zeros(n * limbs, dtype=torch.int64, device=dev)
        ab("%s 2-1 Jive, 2^%d items" % (name, lg),
           lambda: lib.anemoi_jive_compress_k_dev(field, 2, 2, d_in.data_ptr(), d_out.data_ptr(), n, s.cuda_stream), d_out)
h = rng.integers(0, 1 << 60, size=(1 << 17, 4, 4), dtype=np.uint64)
d_in = torch.from_numpy(h.view(np.int64).reshape(-1)).to(dev)
for lg in (13, 14, 15, 16, 17):
    n = 1 << lg
    d_out = torch.zeros(n * 8, dtype=torch.int64, device=dev)
    ab("bn_254 4-3 Jive, 2^%d items" % lg,
       lambda: lib.anemoi_jive_compress_k_dev(2, 4, 2, d_in.data_ptr(), d_out.data_ptr(), n, s.cuda_stream), d_out)
for nmsg_lg, mlen in ((16, 1024), (16, 10240)):
    nmsg = 1 << nmsg_lg
    msgs = torch.from_numpy(rng.integers(0, 256, size=(nmsg, mlen), dtype=np.uint8)).to(dev)
    dig = torch.zeros(nmsg * 4, dtype=torch.int64, device=dev)
    ab("bn_254 4-3 sponge, 2^%d x %d B" % (nmsg_lg, mlen),
       lambda: lib.anemoi_hash_bytes_dev(2, 4, msgs.data_ptr(), mlen, nmsg, dig.data_ptr(), s.cuda_stream), dig)
