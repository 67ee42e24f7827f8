#!/usr/bin/env python3
"""The clock sampler on a HIGH-priority stream (collision-proof: tools/exp_sampler_queue_collision.py) made the FIRST launch
behind its start slow again when the launches follow back to back (tools/measure_cycles.py: config 3 [486.9, 335.3, ...]).
Which arrangement is both collision-proof and harmless?  Each variant: 3 headline launches, one config-3 launch, then the
sampler is started and 3 config-3 launches follow back to back (no host synchronisation in between), as kernel_Mcycles does.

    python tools/exp_sampler_priority_placement.py [reuse]      (reuse: the library's default since the fix -- streams used once and waited for, then kept)
"""
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
    A.warmup("bn_254", 4, 0), A.warmup("bls12_381", 2, 0)

    def cfg3():
        assert A.lib.anemoi_hash_bytes_dev(bn, 4, msgs.data_ptr(), mlen, nmsg, dig.data_ptr(), st.cuda_stream) == 0

    def headline():
        assert A.lib.anemoi_jive_compress_k_dev(bls, 2, 2, d_bls.data_ptr(), o_bls.data_ptr(), n, st.cuda_stream) == 0

    lo, hi = A.ClockSampler.priority_range()
    variants = [("sampler high / stop default, launches back to back", dict(stream_priorities=(hi, lo)), None),
                ("two default-priority streams, back to back", dict(stream_priorities=None), None),
                ("sampler high / stop default, host synchronises after the start", dict(stream_priorities=(hi, lo)), "sync"),
                ("sampler high / stop default, 2 ms of host sleep after the start", dict(stream_priorities=(hi, lo)), "sleep"),
                ("sampler high / stop high", dict(stream_priorities=(hi, hi)), None),
                ("sampler high / stop default, period 200 us", dict(stream_priorities=(hi, lo), period_us=200), None)]
    for rnd in range(2):
        for name, kw, after in variants:
            for _ in range(3):
                headline()
            cfg3()
            torch.cuda.synchronize()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
            cs = A.ClockSampler(dev, reuse_streams="reuse" in sys.argv, **kw)     # default: fresh streams per sampler, as before the fix
            cs.start(st)
            if after == "sync":
                torch.cuda.synchronize()
            elif after == "sleep":
                time.sleep(0.002)
            for a, b in evs:
                a.record(st)
                cfg3()
                b.record(st)
            cs.finish(st)
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in evs]
            print("%-70s %s  groups %d%s" % (name, ["%.1f" % v for v in ms], cs.read()[3], "   <-- slow" if max(ms) > 400 else ""))


if __name__ == "__main__":
    main()
