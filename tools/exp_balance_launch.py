#!/usr/bin/env python3
"""What in front of an underfilled launch makes the dispatcher place it evenly?  (laboratory library: make AB=1)

Config 3's kernel (2 048 single-wavefront workgroups, all resident for 333 ms) takes x 1.46 whenever it follows a launch of a
different big kernel (tools/exp_cfg3_after_other_kernels.py); a launch of the cooperative sponge kernel on 4 096 short messages
in between (2 048 workgroups, 1.5 ms) cures it, capping the workgroups per CU by their LDS request does not (so the imbalance
is between the SIMDs of a CU).  Here: a headline launch, then ONE of several do-nothing launches (k_x_balance: workgroups x
threads, asleep for a few microseconds), then config 3.

    make -C anemoi-rust_amd -j8 AB=1 LIBNAME=libanemoi_ab.so SUFFIX=_ab
    ANEMOI_MI355X_LIB=anemoi-rust_amd/lib/libanemoi_ab.so python tools/exp_balance_launch.py
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
    assert A.is_ab_build()
    A.lib.anemoi_x_balance_dev.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()
    rng = np.random.default_rng(7)
    nmsg, mlen = 1 << 16, 10240
    msgs = torch.from_numpy(rng.integers(0, 256, size=(nmsg, mlen), dtype=np.uint8)).to(dev)
    dig = torch.empty(nmsg * 4, dtype=torch.int64, device=dev)
    bn, bls = A.field_id("bn_254"), A.field_id("bls12_381")
    n = 1 << 20
    d_bls = torch.from_numpy(synth.states("bls12_381", 2, synth.CFG2["seed"], 0, n).view(np.int64).reshape(-1)).to(dev)
    o_bls = torch.empty(n * 6, dtype=torch.int64, device=dev)
    A.set_option("balance_underfilled", 0)

    def timed(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        fn()
        b.record(st)
        torch.cuda.synchronize()
        return a.elapsed_time(b)

    cfg3 = lambda count=nmsg, ln=mlen: timed(lambda: A.lib.anemoi_hash_bytes_dev(bn, 4, msgs.data_ptr(), ln, count, dig.data_ptr(), st.cuda_stream))
    headline = lambda: timed(lambda: A.lib.anemoi_jive_compress_k_dev(bls, 2, 2, d_bls.data_ptr(), o_bls.data_ptr(), n, st.cuda_stream))
    balance = lambda wgs, thr, sl: timed(lambda: A.lib.anemoi_x_balance_dev(wgs, thr, sl, st.cuda_stream))
    print("warm: config 3 %s ms" % ["%.1f" % cfg3() for _ in range(2)])
    variants = [("nothing", None),
                ("cooperative sponge kernel, 4 096 messages of 93 bytes (round 5's cure)", lambda: cfg3(4096, 93)),
                ("do-nothing launch 2 048 x 64 threads, no sleep", lambda: balance(2048, 64, 0)),
                ("do-nothing launch 2 048 x 64, asleep ~20 us", lambda: balance(2048, 64, 6)),
                ("do-nothing launch 4 096 x 64, asleep ~20 us", lambda: balance(4096, 64, 6)),
                ("do-nothing launch 8 192 x 64, asleep ~20 us", lambda: balance(8192, 64, 6)),
                ("do-nothing launch 8 192 x 64, asleep ~200 us", lambda: balance(8192, 64, 60)),
                ("do-nothing launch 16 384 x 64, asleep ~20 us", lambda: balance(16384, 64, 6)),
                ("do-nothing launch 1 024 x 256, asleep ~20 us", lambda: balance(1024, 256, 6)),
                ("do-nothing launch 2 048 x 256, asleep ~20 us", lambda: balance(2048, 256, 6)),
                ("do-nothing launch 256 x 1024, asleep ~20 us", lambda: balance(256, 1024, 6)),
                ("config 3's own kernel on 2 048 x 32 one-byte messages (a launch of its own shape)", lambda: cfg3(nmsg, 1)),
                ("nothing", None)]
    for name, fn in variants:
        h = headline()
        pre = fn() if fn else 0.0
        c = cfg3()
        print("  headline %.1f ms | %-86s %7.3f ms | config 3 %.1f ms%s" % (h, name, pre, c, "   <-- slow" if c > 400 else ""))


if __name__ == "__main__":
    main()
