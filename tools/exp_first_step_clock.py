#!/usr/bin/env python3
"""Why is the FIRST timed step of bench.py slower (103.5, 98.1, 97.0, 97.1 ms)?  The clock each step ran at, from the sampler's
log cut at wall-clock stamps between the steps (anemoi_clock_stamp_dev), after an idle gap like bench.py's (barrier, zero-fill,
synchronise): if kernel time x clock is the same for every step, it is the clock ramping up after the gap.

    python tools/exp_first_step_clock.py [steps=6] [idle_ms=5]
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
import numpy as np
import torch
import anemoi_amd as A
from anemoi_amd import synth


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    idle_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()
    n = 1 << 20
    d_in = torch.from_numpy(synth.states("bls12_381", 2, synth.CFG2["seed"], 0, n).view(np.int64).reshape(-1)).to(dev)
    d_out = torch.zeros(n * 6, dtype=torch.int64, device=dev)
    A.warmup("bls12_381", 2, 0)
    step = lambda: A.lib.anemoi_jive_compress_k_dev(0, 2, 2, d_in.data_ptr(), d_out.data_ptr(), n, st.cuda_stream)
    for label, gap in (("after %.0f ms of idling (bench.py's gap: barrier, zero-fill, synchronise)" % idle_ms, idle_ms), ("after 500 ms of idling", 500.0),
                       ("straight behind three untimed steps (no gap)", 0.0), ("after 1 ms of idling", 1.0), ("after 0.2 ms of idling", 0.2),
                       ("synchronise, then launch at once (sampler started before the untimed steps)", -1.0)):
        cs = A.ClockSampler(dev, period_us=500, max_ms=60000)
        marks = torch.zeros(steps + 1, dtype=torch.int64, device=dev)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        if gap < 0:
            cs.start(st)
        for _ in range(3):
            step()
        if gap:
            torch.cuda.synchronize()
            if gap > 0:
                t_end = time.perf_counter() + gap * 1e-3
                while time.perf_counter() < t_end:
                    pass
        if gap >= 0:
            cs.start(st)
        for i, (a, b) in enumerate(evs):
            A.lib.anemoi_clock_stamp_dev(marks.data_ptr() + 8 * i, st.cuda_stream)
            a.record(st)
            step()
            b.record(st)
        A.lib.anemoi_clock_stamp_dev(marks.data_ptr() + 8 * steps, st.cuda_stream)
        cs.finish(st)
        torch.cuda.synchronize()
        host = cs.buf.cpu().numpy()
        m = marks.cpu().numpy().view(np.uint64)
        print(label)
        for i, (a, b) in enumerate(evs):
            v = [ctypes.c_double(0) for _ in range(3)]
            g = ctypes.c_int(0)
            assert A.lib.anemoi_clock_sampler_read(host.ctypes.data, cs.bytes, int(m[i]), int(m[i + 1]), ctypes.byref(v[0]), ctypes.byref(v[1]),
                                                   ctypes.byref(v[2]), ctypes.byref(g)) == 0
            ms = a.elapsed_time(b)
            print("  step %d: %7.2f ms at mean %.3f / slowest XCD %.3f GHz (%2d sampler groups) -> %.1f / %.1f Mcycles"
                  % (i, ms, v[0].value, v[1].value, g.value, ms * v[0].value, ms * v[1].value))
        cs.read()


if __name__ == "__main__":
    main()
