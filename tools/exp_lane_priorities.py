#!/usr/bin/env python3
"""Do concurrent callers overlap better when every second lane's kernel stream is a high-priority stream (its own set of
hardware queues)?  And do the two kernel streams of a multi-chunk pipeline overlap better when they cannot share a queue?

Child processes (the option is read when a lane is created), ANEMOI_LANE_PRIORITIES = 0 (round 5) and 1:
  * 4 and 8 threads x 3 host-pointer Merkle roots of depth 8 (BLS12-381: eight dependent launches, ~14 ms of kernel latency per
    call) against the same calls from one thread (tests/test_gpu_configs.py::test_concurrent_callers_overlap_on_the_gpu;
    profiles/r05/concurrent_callers_hw_queues.txt: 0.52-0.59 with HIP's 4 queues, 0.38 with GPU_MAX_HW_QUEUES=8);
  * a 2^22-state BLS12-381 Jive batch from pageable host memory (the chunked pipeline: two kernel streams), three times.

    python tools/exp_lane_priorities.py
"""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
    import numpy as np
    import anemoi_amd as A
    rng = np.random.default_rng(9)
    depth, reps = 8, 3
    inst = A.Anemoi("bls12_381", 2)
    for nthreads in (4, 8):
        lvs = [rng.integers(0, 1 << 60, size=(1 << depth, 6), dtype=np.uint64) for _ in range(nthreads)]

        def run(k):
            for _ in range(reps):
                inst.merkle_root(lvs[k], depth)

        ths = [threading.Thread(target=run, args=(k,)) for k in range(nthreads)]   # warm: one lane per thread
        [t.start() for t in ths], [t.join() for t in ths]
        best = (9.0, 0, 0)
        for _ in range(4):
            t0 = time.perf_counter()
            for k in range(nthreads):
                run(k)
            serial = time.perf_counter() - t0
            ths = [threading.Thread(target=run, args=(k,)) for k in range(nthreads)]
            t0 = time.perf_counter()
            [t.start() for t in ths], [t.join() for t in ths]
            conc = time.perf_counter() - t0
            best = min(best, (conc / serial, serial, conc))
        print("  %d threads x %d latency-bound calls: serial %.1f ms, concurrent %.1f ms -> %.2f (perfect %.3f)"
              % (nthreads, reps, best[1] * 1e3, best[2] * 1e3, best[0], 1.0 / nthreads))
    n = 1 << 22
    st = rng.integers(0, 1 << 60, size=(n, 2, 6), dtype=np.uint64)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        inst.compress_batch(st)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("  2^22 states from pageable host memory (chunked pipeline): %s ms" % " ".join("%.1f" % t for t in ts))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child()
    for v in ("0", "1", "0", "1"):
        print("ANEMOI_LANE_PRIORITIES=%s" % v, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, ANEMOI_LANE_PRIORITIES=v), check=True)


if __name__ == "__main__":
    main()
