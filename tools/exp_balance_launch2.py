#!/usr/bin/env python3
"""Second part of tools/exp_balance_launch.py: WHICH do-nothing launch balances WHICH underfilled launch?

Targets: the BN-254 4-3 sponge kernel on 1 024 / 2 048 / 3 072 / 4 096 workgroups (32 messages of 2 KB each per workgroup:
~67 ms when every SIMD holds its even share) and the BLS12-381 Jive kernel on 1 024 / 2 048 / 3 072 workgroups.  Each target
is launched right after a 2^20 Jive launch of ANOTHER field (the disturbing launch), with a do-nothing launch of B single-
wavefront workgroups in between, B in {none, 512, 1 024, 2 048, 4 096, 8 192, the target's own count}.  Reference: the
target launched after itself (steady state).

    ANEMOI_MI355X_LIB=anemoi-rust_amd/lib/libanemoi_ab.so python tools/exp_balance_launch2.py
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
    mlen = 2048
    msgs = torch.from_numpy(rng.integers(0, 256, size=(4096 * 32, mlen), dtype=np.uint8)).to(dev)
    dig = torch.empty(4096 * 32 * 4, dtype=torch.int64, device=dev)
    bn, bls, jub = A.field_id("bn_254"), A.field_id("bls12_381"), A.field_id("jubjub")
    n = 1 << 20
    d_bls = torch.from_numpy(synth.states("bls12_381", 2, synth.CFG2["seed"], 0, n).view(np.int64).reshape(-1)).to(dev)
    d_jub = torch.from_numpy(synth.states("jubjub", 2, 0xA9E30105, 0, n).view(np.int64).reshape(-1)).to(dev)
    o_bls = torch.empty(n * 6, dtype=torch.int64, device=dev)
    o_jub = torch.empty(n * 4, dtype=torch.int64, device=dev)
    A.set_option("balance_underfilled", 0)
    for name in ("coop2d_max", "coop4_max", "coop43_max", "coop2d43_max", "coop_sponge_max"):
        A.set_option(name, 0)          # the lane-private kernels at every size

    def timed(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        fn()
        b.record(st)
        torch.cuda.synchronize()
        return a.elapsed_time(b)

    sponge = lambda wgs: timed(lambda: A.lib.anemoi_hash_bytes_dev(bn, 4, msgs.data_ptr(), mlen, wgs * 32, dig.data_ptr(), st.cuda_stream))
    jive_bls = lambda wgs: timed(lambda: A.lib.anemoi_jive_compress_k_dev(bls, 2, 2, d_bls.data_ptr(), o_bls.data_ptr(), wgs * 64, st.cuda_stream))
    jive_jub_full = lambda: timed(lambda: A.lib.anemoi_jive_compress_k_dev(jub, 2, 2, d_jub.data_ptr(), o_jub.data_ptr(), n, st.cuda_stream))
    jive_bls_full = lambda: timed(lambda: A.lib.anemoi_jive_compress_k_dev(bls, 2, 2, d_bls.data_ptr(), o_bls.data_ptr(), n, st.cuda_stream))
    balance = lambda wgs: timed(lambda: A.lib.anemoi_x_balance_dev(wgs, 64, 6, st.cuda_stream))
    targets = [("sponge bn_254 4-3", sponge, jive_bls_full, (1024, 2048, 3072, 4096)),
               ("jive bls12_381 2-1", jive_bls, jive_jub_full, (512, 1024, 2048, 3072))]
    for tname, target, disturb, sizes in targets:
        for wgs in sizes:
            target(wgs)
            steady = min(target(wgs) for _ in range(2))
            row = []
            for b in (None, 512, 1024, 2048, 4096, 8192, wgs):
                disturb()
                if b:
                    balance(b)
                row.append((b, target(wgs)))
            print("%-20s %5d workgroups: after itself %7.2f ms | after another kernel's 2^20 launch + a do-nothing launch of B workgroups: %s"
                  % (tname, wgs, steady, "  ".join("B=%s %.2f%s" % (b if b else "none", t, "*" if t > 1.15 * steady else "") for b, t in row)))


if __name__ == "__main__":
    main()
