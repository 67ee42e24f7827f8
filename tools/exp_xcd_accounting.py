#!/usr/bin/env python3
"""Where do the 1-3 % between the MEAN and the SLOWEST sampled XCD clock go?  (round 5's review, item 5)

bench.py's box-independent figure is kernel time x the SLOWEST sampled XCD clock (228.5 +- 1 % Mcycles on every box);
with the mean clock the same launch costs 231-235.  Workgroups are dealt to the XCDs statically (2 048 of 16 384 each),
the XCDs hold different clocks -- and yet handing the blocks out dynamically (k_jive_queue, k_jive_ticket) changed nothing
(profiles/r05/jive_work_queue_not_adopted.txt).  This tool asks the launch itself: in a `make AB=1` library every block of
the headline kernel records, per XCD (XCC_ID), that it ran, when it started and ended (100 MHz wall clock) and how many
wall ticks and shader cycles it took (anemoi_kernels.h: XcdAcct) -- under the shipped static dealing (k_jive_acct), the
work queue (3 072 resident workgroups) and the tickets (x 1.25 workgroups).  Per XCD it prints

    blocks      how many 64-state blocks the XCD worked
    GHz         cycles / wall ticks of its blocks = the clock it held under its own work, seen by the work's wavefronts
    kcyc/blk    shader cycles per block (does a block cost the same everywhere?)
    ms/blk      wall time per block
    start, end  first block's start / last block's end, ms after the launch's first start

and, per launch, the spread of the XCDs' ends and what the time would have been had every XCD run at the mean clock.

    make -C anemoi-rust_amd -j8 AB=1 LIBNAME=libanemoi_ab.so SUFFIX=_ab
    ANEMOI_MI355X_LIB=anemoi-rust_amd/lib/libanemoi_ab.so python tools/exp_xcd_accounting.py [--log2 20] [--rounds 3]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
import numpy as np
import torch
import anemoi_amd as A
from anemoi_amd import synth


def main():
    lg = int(sys.argv[sys.argv.index("--log2") + 1]) if "--log2" in sys.argv else 20
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
    assert A.is_ab_build(), "needs a laboratory library: make AB=1 ... and ANEMOI_MI355X_LIB (see the docstring)"
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    A.lib.anemoi_x_jive_acct_dev.argtypes = [ctypes.c_int, vp, vp, sz, vp, ctypes.c_uint, vp, vp]
    A.lib.anemoi_x_jive_acct_dev.restype = ctypes.c_int
    dev = torch.device("cuda", 0)
    work = torch.cuda.current_stream()
    assert A.lib.anemoi_init(0, 0, 2) == 0
    A.warmup("bls12_381", 2, 0)
    n = 1 << lg
    blocks = n // 64
    host = synth.states("bls12_381", 2, synth.CFG2["seed"], 0, n)
    d_in = torch.from_numpy(host.view(np.int64).reshape(-1)).to(dev)
    d_ref = torch.zeros(n * 6, dtype=torch.int64, device=dev)
    d_out = torch.zeros(n * 6, dtype=torch.int64, device=dev)
    q = torch.zeros(4, dtype=torch.int32, device=dev)
    assert A.lib.anemoi_jive_compress_k_dev(0, 2, 2, d_in.data_ptr(), d_ref.data_ptr(), n, work.cuda_stream) == 0
    fresh = np.zeros((8, 5), dtype=np.uint64)
    fresh[:, 1] = np.uint64((1 << 64) - 1)                     # first_start: minimum over the blocks
    variants = [("static dealing (the shipped kernel's: block = blockIdx.x)", 0),
                ("work queue, 3 072 resident workgroups", 3072),
                ("tickets, 1.25 x as many workgroups as blocks", int(blocks * 1.25))]
    print("2^%d BLS12-381 Anemoi-2-1 compressions = %d blocks of 64; per XCD, per launch (round r of %d, interleaved variants)" % (lg, blocks, rounds))
    summary = {name: [] for name, _ in variants}
    for rnd in range(rounds):
        for name, wgs in variants:
            acct = torch.from_numpy(fresh.view(np.int64).copy()).to(dev)
            d_out.zero_()
            cs = A.ClockSampler(dev)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cs.start(work)
            a.record(work)
            assert A.lib.anemoi_x_jive_acct_dev(0, d_in.data_ptr(), d_out.data_ptr(), n, q.data_ptr(), wgs, acct.data_ptr(), work.cuda_stream) == 0
            b.record(work)
            cs.finish(work)
            torch.cuda.synchronize()
            assert torch.equal(d_out, d_ref), "the accounted kernel computes something else"
            ms = a.elapsed_time(b)
            s_mean, s_lo, s_hi, _ = cs.read()
            r = acct.cpu().numpy().view(np.uint64).astype(np.float64).reshape(8, 5)
            nb, first, last, ticks, cyc = r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4]
            assert nb.sum() == blocks, nb
            t0 = first.min()
            ghz = cyc / ticks * 0.1                                   # 100 MHz ticks -> GHz
            end_ms = (last - t0) * 1e-5
            print("\n%s -- round %d: %.3f ms by HIP events; sampler beside it: mean %.4f, slowest %.4f, fastest %.4f GHz" % (name, rnd, ms, s_mean, s_lo, s_hi))
            print("  XCD  blocks     GHz   kcyc/blk   ms/blk   start ms    end ms")
            for x in range(8):
                print("  %3d  %6d  %6.4f  %9.1f  %7.3f  %9.3f  %8.3f" % (x, nb[x], ghz[x], cyc[x] / nb[x] / 1e3, ticks[x] / nb[x] * 1e-5,
                                                                         (first[x] - t0) * 1e-5, end_ms[x]))
            spread = (end_ms.max() - end_ms.min()) / end_ms.max()
            wmean = (ghz * nb).sum() / nb.sum()
            corr = float(np.corrcoef(ghz, nb)[0, 1]) if nb.std() > 0 else float("nan")
            corr_end = float(np.corrcoef(ghz, end_ms)[0, 1])
            print("  ends: first XCD done at %.3f ms, last at %.3f ms (spread %.2f %% of the launch); clock under the work: mean %.4f, slowest %.4f, "
                  "fastest %.4f GHz (%.2f %% apart); cycles per block differ by %.2f %% across XCDs"
                  % (end_ms.min(), end_ms.max(), 100 * spread, wmean, ghz.min(), ghz.max(), 100 * (ghz.max() / ghz.min() - 1),
                     100 * ((cyc / nb).max() / (cyc / nb).min() - 1)))
            print("  correlation(clock, blocks taken) = %s; correlation(clock, end) = %.2f; launch x slowest work clock = %.2f Mcycles, x mean = %.2f"
                  % ("%.2f" % corr if corr == corr else "n/a (equal shares)", corr_end, end_ms.max() * ghz.min(), end_ms.max() * wmean))
            summary[name].append((ms, end_ms.max(), 100 * spread, ghz.min(), wmean, ghz.max(), nb.min(), nb.max(), (cyc / nb).mean() / 1e3))
    print("\nMedians over %d rounds" % rounds)
    print("  %-62s  %8s  %8s  %7s  %-24s  %-13s  %s" % ("variant", "ms", "last end", "spread", "work clock min/mean/max", "blocks/XCD", "kcyc/blk"))
    for name, _ in variants:
        v = np.median(np.array(summary[name]), axis=0)
        print("  %-62s  %8.3f  %8.3f  %6.2f%%  %.4f / %.4f / %.4f  %5d .. %5d  %8.1f" % (name, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]))


if __name__ == "__main__":
    main()
