#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point (what the Rust shim calls):
anemoi_jive_compress_batch on 2^20 BLS12-381 states in pageable host memory."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
import numpy as np
import bench
import anemoi_amd as A

n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
st = bench.synth_states(n, 5)
inst = A.Anemoi("bls12_381", 2)
inst.compress_batch(st[:1024])
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    out = inst.compress_batch(st)
    ts.append(time.perf_counter() - t0)
t = sorted(ts)[1]
print("host-pointer anemoi_jive_compress_batch: %d items in %.1f ms -> %.2f M compress/s (PCIe + alloc inclusive)"
      % (n, t * 1e3, n / t / 1e6))
