#!/usr/bin/env python3
"""The clock the chip holds WHILE the headline kernel runs (anemoi_clock_sampler_*): kernel time x clock should be the
same number of cycles on every box if the clock is what makes a box fast or slow."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
import numpy as np, torch
import anemoi_amd as A
from anemoi_amd import synth
dev = torch.device("cuda", 0)
n = 1 << 20
host = synth.states("bls12_381", 2, synth.CFG2["seed"], 0, n)
d_in = torch.from_numpy(host.view(np.int64).reshape(-1)).to(dev)
d_out = torch.zeros(n * 6, dtype=torch.int64, device=dev)
assert A.lib.anemoi_init(0, 0, 2) == 0
work, side, third = torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()
nb = A.lib.anemoi_clock_sampler_bytes()
buf = torch.zeros(nb, dtype=torch.uint8, device=dev)
stamps = torch.zeros(2, dtype=torch.int64, device=dev)
for rep in range(3):
    for _ in range(2):
        assert A.lib.anemoi_jive_compress_k_dev(0, 2, 2, d_in.data_ptr(), d_out.data_ptr(), n, work.cuda_stream) == 0
    torch.cuda.synchronize()
    assert A.lib.anemoi_clock_sampler_start_dev(buf.data_ptr(), nb, 2000, 20000, side.cuda_stream) == 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    assert A.lib.anemoi_clock_stamp_dev(stamps.data_ptr(), work.cuda_stream) == 0
    a.record(work)
    K = 5
    for _ in range(K):
        assert A.lib.anemoi_jive_compress_k_dev(0, 2, 2, d_in.data_ptr(), d_out.data_ptr(), n, work.cuda_stream) == 0
    b.record(work)
    assert A.lib.anemoi_clock_stamp_dev(stamps.data_ptr() + 8, work.cuda_stream) == 0
    work.synchronize()
    assert A.lib.anemoi_clock_sampler_stop_dev(buf.data_ptr(), third.cuda_stream) == 0
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / K
    h = buf.cpu().numpy()
    st = stamps.cpu().numpy().view(np.uint64)
    v = [ctypes.c_double(0) for _ in range(3)]
    g = ctypes.c_int(0)
    assert A.lib.anemoi_clock_sampler_read(h.ctypes.data, nb, int(st[0]), int(st[1]), ctypes.byref(v[0]), ctypes.byref(v[1]), ctypes.byref(v[2]), ctypes.byref(g)) == 0
    print("kernel %.2f ms | wall between stamps %.2f ms | clock during it: mean %.4f GHz (min %.4f, max %.4f over %d sampler workgroups) | "
          "kernel Mcycles at the mean / slowest clock %.2f / %.2f | lane-mad fraction at the mean / slowest clock %.4f / %.4f"
          % (ms, (int(st[1]) - int(st[0])) / 1e5, v[0].value, v[1].value, v[2].value, g.value, ms * v[0].value, ms * v[1].value,
             2724592 * n / (ms * 1e-3) / (1024 * 16 * v[0].value * 1e9), 2724592 * n / (ms * 1e-3) / (1024 * 16 * v[1].value * 1e9)))
