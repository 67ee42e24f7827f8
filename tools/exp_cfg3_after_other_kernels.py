#!/usr/bin/env python3
"""When is a launch of config 3's kernel (k_sponge_pair<bn_254, true>: 2 048 single-wavefront workgroups that all start
at once and stay ~333 ms) placed badly (x 1.46: all of them on two thirds of the CUs)?

Round 5 tied it to "the first big launch of a process" and cured it with anemoi_warmup on the two boxes tried.  Round 6's
cycles tool saw [487, 335, 333] ms for the first TIMED launch although anemoi_warmup and an untimed launch of the same
kernel had run -- after 2^20-state launches of the headline kernel.  This tool times every launch of a scripted sequence
in one process:

    cfg3 x 3 | headline x 2, cfg3 x 2 | jubjub 2^20 x 2, cfg3 x 2 | small cfg3-kernel launch (4 096 messages), cfg3 x 2
    | headline x 2, small launch, cfg3 x 2 | sampler beside: cfg3 x 2 | headline x 2, sampler beside: cfg3 x 2

    python tools/exp_cfg3_after_other_kernels.py [api_warmup] [no_balance] [sampler_default_priority]
"""
import os
import sys

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
    bn, bls, jub = A.field_id("bn_254"), A.field_id("bls12_381"), A.field_id("jubjub")
    n = 1 << 20
    d_bls = torch.from_numpy(synth.states("bls12_381", 2, synth.CFG2["seed"], 0, n).view(np.int64).reshape(-1)).to(dev)
    d_jub = torch.from_numpy(synth.states("jubjub", 2, 0xA9E30105, 0, n).view(np.int64).reshape(-1)).to(dev)
    o_bls = torch.empty(n * 6, dtype=torch.int64, device=dev)
    o_jub = torch.empty(n * 4, dtype=torch.int64, device=dev)
    if "api_warmup" in sys.argv:
        A.warmup("bn_254", 4, 0), A.warmup("bls12_381", 2, 0), A.warmup("jubjub", 2, 0)
        print("anemoi_warmup of the three instances first")

    def timed(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        fn()
        b.record(st)
        torch.cuda.synchronize()
        return a.elapsed_time(b)

    def cfg3(count=nmsg, ln=mlen):
        return timed(lambda: A.lib.anemoi_hash_bytes_dev(bn, 4, msgs.data_ptr(), ln, count, dig.data_ptr(), st.cuda_stream))

    def headline():
        return timed(lambda: A.lib.anemoi_jive_compress_k_dev(bls, 2, 2, d_bls.data_ptr(), o_bls.data_ptr(), n, st.cuda_stream))

    def jubjub():
        return timed(lambda: A.lib.anemoi_jive_compress_k_dev(jub, 2, 2, d_jub.data_ptr(), o_jub.data_ptr(), n, st.cuda_stream))

    def show(what, vals):
        print("  %-66s %s%s" % (what, ["%.1f" % v for v in vals], "   <-- slow" if max(vals) > 400 else ""))

    if "no_balance" in sys.argv:
        A.set_option("balance_underfilled", 0)
        print("option balance_underfilled = 0: no do-nothing launch in front of underfilled launches (rounds 1-5)")
    show("config 3 x 3 (the process's first launches)", [cfg3() for _ in range(3)])
    show("headline (k_jive<bls12_381>, 2^20) x 2", [headline() for _ in range(2)])
    show("  then config 3 x 2", [cfg3() for _ in range(2)])
    show("k_jive<jubjub>, 2^20 x 2", [jubjub() for _ in range(2)])
    show("  then config 3 x 2", [cfg3() for _ in range(2)])
    show("the config-3 kernel on 4 096 messages of 93 bytes", [cfg3(4096, 93)])
    show("  then config 3 x 2", [cfg3() for _ in range(2)])
    show("headline x 2", [headline() for _ in range(2)])
    show("  then the small launch", [cfg3(4096, 93)])
    show("  then config 3 x 2", [cfg3() for _ in range(2)])
    for pre in (False, True):
        if pre:
            show("headline x 2", [headline() for _ in range(2)])
        cs = A.ClockSampler(dev, stream_priorities=None if "sampler_default_priority" in sys.argv else "auto")
        cs.start(st)
        vals = [cfg3() for _ in range(2)]
        cs.finish(st)
        torch.cuda.synchronize()
        show("%sclock sampler beside: config 3 x 2 (sampled clock min %.3f GHz)" % ("  then " if pre else "", cs.read()[1]), vals)
    show("headline x 1, config 3 x 1, alternating three times", [f() for _ in range(3) for f in (headline, cfg3)])
    for rep in range(4):      # what tools/measure_cycles.py does: headline launches, ONE config-3 launch, then the sampler and config 3
        vals = [headline() for _ in range(3)] + [cfg3()]
        cs = A.ClockSampler(dev, stream_priorities=None if "sampler_default_priority" in sys.argv else "auto")
        cs.start(st)
        vals += [cfg3() for _ in range(2)]
        cs.finish(st)
        torch.cuda.synchronize()
        show("headline x 3, config 3, then sampler beside config 3 x 2", vals[3:])


if __name__ == "__main__":
    main()
