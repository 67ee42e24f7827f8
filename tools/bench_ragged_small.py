#!/usr/bin/env python3
"""A SMALL device-resident ragged batch (a service's handful of requests of different lengths): which kernel hashes it?
Before k_sponge_ragged_coop every ragged batch took the lane-private ragged kernels (one permutation per 2.25 ms on
Jubjub whatever the batch size); now batches up to the cut-offs of the equal-length sponge take the latency kernels
(two-row fold up to one wavefront per SIMD, the scan up to four).  Both routings are timed in one process through the
cut-off options, each the best of five launches, digests compared.
    python tools/bench_ragged_small.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
import numpy as np
import torch
import anemoi_amd as A

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
rng = np.random.default_rng(12)
LANE_PRIVATE = dict(coop2d_max=0, coop2d43_max=0, coop_sponge_max=0)


def timed(call, reps=5):
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        assert call() == 0
        b.record(stream)
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


print("%-10s %-4s %6s %-22s %12s %14s %7s" % ("field", "", "n", "lengths (bytes)", "latency ms", "lane-private ms", "ratio"))
for field, width in (("jubjub", 2), ("bls12_381", 2), ("bn_254", 4), ("bls12_381", 4)):
    fid, inst = A.field_id(field), A.Anemoi(field, width)
    L = inst.limbs
    A.warmup(field, width)
    for n, lo, hi in ((1, 300, 301), (16, 0, 600), (256, 0, 600), (1024, 0, 600), (2048, 0, 600), (4096, 0, 600), (256, 0, 6000)):
        lens = rng.integers(lo, hi, size=n)
        offs = np.zeros(n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lens, dtype=np.uint64)
        blob = rng.integers(0, 256, size=int(offs[-1]) + 1, dtype=np.uint8)
        d_blob, d_offs = torch.from_numpy(blob).to(dev), torch.from_numpy(offs.view(np.int64)).to(dev)
        d_out = [torch.zeros(n * L, dtype=torch.int64, device=dev) for _ in range(2)]
        d_scr = torch.empty(A.lib.anemoi_ragged_scratch_bytes(n), dtype=torch.uint8, device=dev)

        def call(o):
            return A.lib.anemoi_hash_bytes_ragged_bucketed_dev(fid, width, d_blob.data_ptr(), d_blob.numel(), d_offs.data_ptr(), n, o.data_ptr(),
                                                               d_scr.data_ptr(), d_scr.numel(), stream.cuda_stream)
        t_new = timed(lambda: call(d_out[0]))
        with A.options(**LANE_PRIVATE):
            t_old = timed(lambda: call(d_out[1]))
        assert torch.equal(d_out[0], d_out[1])
        print("%-10s %d-%d %6d %-22s %12.2f %14.2f %7.2f" % (field, width, width - 1, n, "%d ... %d" % (lo, hi - 1), t_new, t_old, t_old / t_new))
