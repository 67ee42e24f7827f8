#!/usr/bin/env python3
"""The box-independent cost -- kernel time x the slowest sampled XCD clock, in millions of shader cycles -- of the three
kernels the BASELINE configs spend their time in, measured with the library's own clock sampler beside the work
(anemoi_amd.kernel_Mcycles; bench.py's alu.kernel_Mcycles_slowest_xcd for any kernel):

    headline   k_jive<bls12_381, 2, 2>      2^20 BLS12-381 Anemoi-2-1 compressions   (config 2, bench.py's step)
    cfg3       k_sponge_pair<bn_254, true>   2^16 messages of 10 240 bytes           (config 3)
    cfg5_top   k_jive<jubjub, 2, 2>          2^20 Jubjub merges: the widest level of one GPU's depth-21 subtree (config 5)

tests/test_gpu_cycles.py holds the budgets (a slower BUILD fails there on any box; a slower box does not);
profiles/rNN/cycles_budget.json is this tool's output on the round's final sources.

    python tools/measure_cycles.py [--reps 3] [--out gpurun_out/cycles.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "anemoi-rust_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch


def workloads(A, dev):
    """name -> (enqueue(stream), items per repetition, description).  Inputs resident in HBM; seeded."""
    from anemoi_amd import synth
    out = {}

    def jive(field, limbs, n, seed):
        st = synth.states(field, 2, seed, 0, n)
        d_in = torch.from_numpy(st.view(np.int64).reshape(-1)).to(dev)
        d_out = torch.empty(n * limbs, dtype=torch.int64, device=dev)
        fid = A.field_id(field)

        def enqueue(stream):
            rc = A.lib.anemoi_jive_compress_k_dev(fid, 2, 2, d_in.data_ptr(), d_out.data_ptr(), n, stream.cuda_stream)
            assert rc == 0, rc
        enqueue.keep = (d_in, d_out)
        return enqueue

    out["headline"] = (jive("bls12_381", 6, 1 << 20, synth.CFG2["seed"]), 1 << 20,
                       "k_jive<bls12_381,2,2>: 2^20 Anemoi-2-1 compressions (config 2)")
    out["cfg5_top"] = (jive("jubjub", 4, 1 << 20, 0xA9E30105), 1 << 20,
                       "k_jive<jubjub,2,2>: 2^20 merges, the widest level of a depth-21 subtree (config 5)")
    nmsg, mlen = 1 << 16, 10240
    rng = np.random.default_rng(0xA9E30103)
    msgs = torch.from_numpy(rng.integers(0, 256, size=(nmsg, mlen), dtype=np.uint8)).to(dev)
    dig = torch.empty(nmsg * 4, dtype=torch.int64, device=dev)
    bn = A.field_id("bn_254")

    def sponge(stream):
        rc = A.lib.anemoi_hash_bytes_dev(bn, 4, msgs.data_ptr(), mlen, nmsg, dig.data_ptr(), stream.cuda_stream)
        assert rc == 0, rc
    sponge.keep = (msgs, dig)
    out["cfg3"] = (sponge, nmsg, "k_sponge_pair<bn_254,true>: 2^16 messages of 10 240 bytes (config 3)")
    return out


def measure(A, dev, names=("headline", "cfg3", "cfg5_top"), reps=3, best=False):
    from anemoi_amd import buildinfo
    A.warmup("bls12_381", 2, 0), A.warmup("jubjub", 2, 0), A.warmup("bn_254", 4, 0)
    res = {}
    wl = workloads(A, dev)
    for name in names:
        enqueue, items, what = wl[name]
        ms, ghz, mcyc = A.kernel_Mcycles(dev, enqueue, reps=reps, warmup=1, best=best)
        res[name] = {"what": what, "items": items, "ms": ms, "ms_each": list(A.kernel_Mcycles.last_each_ms), "clock_GHz_slowest_xcd": ghz, "Mcycles_slowest_xcd": mcyc,
                     "items_per_s": items / (ms * 1e-3)}
    return {"csrc_sha256": buildinfo.csrc_sha256(), "reps": reps, "best_of_reps": best, "kernels": res,
            "note": "ms = HIP events on the work's stream, mean (or, best_of_reps, the fastest) of `reps`; clock = slowest of the XCD clocks sampled beside the work "
                    "(anemoi_clock_sampler_*); Mcycles = ms x clock: what a BUILD costs whatever the box"}


def main():
    import anemoi_amd as A
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
    res = measure(A, torch.device("cuda", 0), reps=reps, best="--best" in sys.argv)
    text = json.dumps(res, indent=1)
    print(text)
    if "--out" in sys.argv:
        with open(sys.argv[sys.argv.index("--out") + 1], "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
