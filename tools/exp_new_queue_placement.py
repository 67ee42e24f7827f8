#!/usr/bin/env python3
"""Is the x 1.47 of a "first big launch" tied to the FIRST USE OF A HARDWARE QUEUE rather than to the process's start?

Seen in round 6 (gpurun_out/r06/cycles_s1.json): config 3's kernel measured with the clock sampler beside it took
[488.6, 335.8, 333.6] ms although anemoi_warmup AND one untimed launch of the same kernel had run before -- round 5's cure
for the first-launch placement effect was in place.  What was new in front of the slow launch: the
sampler's streams, used for the first time.  This tool brings config 3's kernel to its steady state and then, before each
further launch, does ONE thing:

    nothing | a trivial kernel on a NEW stream (a queue's first use) | the same on that stream again (the queue exists)
    | the clock sampler started on new streams | the clock sampler again (its streams exist) | 300 ms of idling

and prints the launch's kernel time.  If a new queue's first use makes the next big dispatch slow, the cure is to create
the library's streams (and the sampler's) before any timed work, not to launch the kernel once more.

    python tools/exp_new_queue_placement.py
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
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()
    rng = np.random.default_rng(7)
    nmsg, mlen = 1 << 16, 10240
    msgs = torch.from_numpy(rng.integers(0, 256, size=(nmsg, mlen), dtype=np.uint8)).to(dev)
    dig = torch.empty(nmsg * 4, dtype=torch.int64, device=dev)
    bn = A.field_id("bn_254")
    A.warmup("bn_254", 4, 0)

    def cfg3():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        assert A.lib.anemoi_hash_bytes_dev(bn, 4, msgs.data_ptr(), mlen, nmsg, dig.data_ptr(), st.cuda_stream) == 0
        b.record(st)
        torch.cuda.synchronize()
        return a.elapsed_time(b)

    print("after anemoi_warmup, three launches: %s" % ["%.1f" % cfg3() for _ in range(3)])
    side = {}

    def new_stream_kernel(key):
        fresh = key not in side
        if fresh:
            side[key] = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side[key]):
            torch.zeros(64, device=dev).add_(1)
        side[key].synchronize()
        return "a trivial kernel on %s" % ("a NEW stream" if fresh else "that stream again")

    samplers = []

    def sampler_beside():
        cs = A.ClockSampler(dev)
        cs.start(st)
        samplers.append(cs)
        return "clock sampler started (%s)" % ("its first use in the process" if len(samplers) == 1 else "its streams exist")

    def sampler_stop():
        cs = samplers[-1]
        cs.finish(st)
        torch.cuda.synchronize()

    steps = [("nothing", lambda: "nothing", None), ("new stream", lambda: new_stream_kernel("a"), None),
             ("same stream", lambda: new_stream_kernel("a"), None), ("another new stream", lambda: new_stream_kernel("b"), None),
             ("nothing", lambda: "nothing", None),
             ("sampler", sampler_beside, sampler_stop), ("sampler again", sampler_beside, sampler_stop),
             ("idle", lambda: (time.sleep(0.3), "300 ms of idling")[1], None), ("nothing", lambda: "nothing", None)]
    for _, before, after in steps:
        what = before()
        ms = cfg3()
        if after:
            after()
        print("  before the launch: %-62s -> %.1f ms%s" % (what, ms, "   <-- slow" if ms > 400 else ""))
    # ... and the same in the other order for the sampler: fresh streams of ITS OWN first, then the sampler
    for k in ("c", "d", "e"):
        new_stream_kernel(k)
    print("three more new streams used once each, then: %.1f ms, %.1f ms" % (cfg3(), cfg3()))


if __name__ == "__main__":
    main()
