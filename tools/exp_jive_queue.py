#!/usr/bin/env python3
"""Static dealing of workgroups against a work queue (k_jive_queue: a recorded negative, compiled into `make AB=1` libraries
only), same process, interleaved; the clock sampler beside both.
    make -C anemoi-rust_amd -j8 AB=1 LIBNAME=libanemoi_ab.so && ANEMOI_MI355X_LIB=anemoi-rust_amd/lib/libanemoi_ab.so python tools/exp_jive_queue.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
import numpy as np, torch
import anemoi_amd as A
from anemoi_amd import synth
vp, sz = ctypes.c_void_p, ctypes.c_size_t
A.lib.anemoi_x_jive_queue_dev.argtypes = [ctypes.c_int, vp, vp, sz, vp, ctypes.c_uint, vp]
dev = torch.device("cuda", 0)
work = torch.cuda.current_stream()
assert A.lib.anemoi_init(0, 0, 2) == 0
q = torch.zeros(4, dtype=torch.int32, device=dev)
for lg in (20, 21):
    n = 1 << lg
    host = synth.states("bls12_381", 2, synth.CFG2["seed"], 0, n)
    d_in = torch.from_numpy(host.view(np.int64).reshape(-1)).to(dev)
    d_a = torch.zeros(n * 6, dtype=torch.int64, device=dev)
    d_b = torch.zeros(n * 6, dtype=torch.int64, device=dev)
    def static():
        assert A.lib.anemoi_jive_compress_k_dev(0, 2, 2, d_in.data_ptr(), d_a.data_ptr(), n, work.cuda_stream) == 0
    variants = [("static", static)]
    for wgs in (3072, 6144):
        def queue(wgs=wgs):
            assert A.lib.anemoi_x_jive_queue_dev(0, d_in.data_ptr(), d_b.data_ptr(), n, q.data_ptr(), wgs, work.cuda_stream) == 0
        variants.append(("queue, %d workgroups" % wgs, queue))
    for over in (1.0, 1.0625, 1.125, 1.25):     # tickets: one block per workgroup, `over` x as many workgroups as blocks
        wgs = int(n // 64 * over)
        def ticket(wgs=wgs):
            assert A.lib.anemoi_x_jive_queue_dev(0, d_in.data_ptr(), d_b.data_ptr(), n, q.data_ptr(), wgs, work.cuda_stream) == 0
        variants.append(("tickets, x %.4f" % over, ticket))
    for name, fn in variants:
        fn()
    torch.cuda.synchronize()
    assert torch.equal(d_a, d_b), "the queue kernel computes something else"
    res = {name: [] for name, _ in variants}
    for rnd in range(4):
        for name, fn in variants:
            cs = A.ClockSampler(dev)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cs.start(work)
            a.record(work)
            for _ in range(3):
                fn()
            b.record(work)
            cs.finish(work)
            torch.cuda.synchronize()
            mean, lo, hi, g = cs.read()
            res[name].append((a.elapsed_time(b) / 3, mean, lo))
    for name, _ in variants:
        r = sorted(res[name])[len(res[name]) // 2]
        print("2^%d  %-24s median %8.3f ms  -> %6.3f M/s | clock mean %.4f slowest XCD %.4f | Mcycles at mean / slowest %.2f / %.2f"
              % (lg, name, r[0], n / r[0] / 1e3, r[1], r[2], r[0] * r[1], r[0] * r[2]))
