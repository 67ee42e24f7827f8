#!/usr/bin/env python3
"""Does the clock sampler's stream share a hardware queue with the work's stream -- and what does that cost?

HIP multiplexes the streams of a process onto GPU_MAX_HW_QUEUES (default 4) hardware queues and runs the kernels of
streams that share a queue one after the other (profiles/r05/concurrent_callers_hw_queues.txt).  The sampler kernel never
ends by itself; if its stream shares a queue with the work's stream the work waits until the sampler's log is full
(4 096 periods) and then runs WITHOUT a sampler -- "the clock sampler took no sample beside the work", seen in round 6 in
some test orders.  Here: N samplers in a row, each on freshly taken torch streams, beside 25 ms of work on the default
stream; per sampler the wall time of the whole piece and the number of sampler workgroups that saw the work --
with two default-priority streams (round 5), with the sampler alone on a default-priority stream, with the sampler on a
high-priority stream (round 6), and with round 5's streams while other code takes a stream in between.

    python tools/exp_sampler_queue_collision.py [n=24]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
import numpy as np
import torch
import anemoi_amd as A


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    dev = torch.device("cuda", 0)
    work = torch.cuda.current_stream()
    print("torch.cuda.Stream.priority_range():", A.ClockSampler.priority_range(), "(lowest, highest)")
    big = np.random.default_rng(6).integers(0, 1 << 60, size=(1 << 17, 2, 4), dtype=np.uint64)
    d_in = torch.from_numpy(big.view(np.int64).reshape(-1)).to(dev)
    d_out = torch.zeros((1 << 17) * 4, dtype=torch.int64, device=dev)
    jub = A.field_id("jubjub")
    A.warmup("jubjub", 2, 0)
    lo, hi = A.ClockSampler.priority_range()
    for label, prio in (("two default-priority streams (round 5)", None),
                        ("sampler on a default-priority stream, stop on a high-priority one", (lo, hi)),
                        ("sampler on a HIGH-priority stream, stop on a default one (round 6)", "auto"),
                        ("two default-priority streams, an odd stream taken before each", "odd")):
        res = []
        for i in range(n):
            if prio == "odd":
                keep = torch.cuda.Stream(dev)          # what other code in a process does: the round-robin shifts by one
                with torch.cuda.stream(keep):
                    torch.zeros(8, device=dev)
            cs = A.ClockSampler(dev, period_us=100, max_ms=20000, stream_priorities=None if prio == "odd" else prio, reuse_streams=False)   # a full log = 0.41 s
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cs.start(work)
            for _ in range(6):
                assert A.lib.anemoi_jive_compress_k_dev(jub, 2, 2, d_in.data_ptr(), d_out.data_ptr(), 1 << 17, work.cuda_stream) == 0
            cs.finish(work)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            res.append((ms, cs.read()[3]))
        bad = [i for i, (ms, g) in enumerate(res) if g < 12]
        print("%-70s %d samplers: %d without the work in their log %s; wall ms per piece: %s"
              % (label, n, len(bad), bad, " ".join("%.0f" % ms for ms, _ in res)))


if __name__ == "__main__":
    main()
