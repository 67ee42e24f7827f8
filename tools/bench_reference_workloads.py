#!/usr/bin/env python3
"""The reference's criterion workloads (benches/<field>.rs, identical for the 7 fields) on the GPU:

    compress/2-1, compress/4-3, compress_k(4)/4-3      on [Felt::one(); STATE_WIDTH]   (benches/bls12_381.rs:12-37)
    hash/2-1, hash/4-3                                  on 10 * 1024 random bytes       (benches/bls12_381.rs:39-59)

criterion times ONE call; a GPU serves batches, so both are reported: the latency of a batch of one
(what `Jive::compress` through the shim costs) and the amortised time per item in a large batch.
The reference's published CPU numbers (README.md:73-85, i7-9750H, one thread) are printed beside
them where they exist.  Inputs resident in HBM, HIP events on the launch stream.
    python tools/bench_reference_workloads.py [--fields bls12_377,vesta] [--big 18]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

FIELDS = ["bls12_381", "bls12_377", "bn_254", "ed_on_bls12_377", "jubjub", "pallas", "vesta"]
LIMBS = [6, 6, 4, 4, 4, 4, 4]
README_US = {  # reference README.md:77-78,84-85 (microseconds per call, CPU)
    ("bls12_377", "compress/2-1"): 429.61, ("bls12_377", "compress/4-3"): 485.99,
    ("vesta", "compress/2-1"): 129.48, ("vesta", "compress/4-3"): 176.58,
    ("bls12_377", "hash10KB/2-1"): 85369.0, ("bls12_377", "hash10KB/4-3"): 35937.0,
    ("vesta", "hash10KB/2-1"): 44448.0, ("vesta", "hash10KB/4-3"): 20307.0,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fields", default=",".join(FIELDS))
    ap.add_argument("--big", type=int, default=18, help="log2 of the large compress batch")
    ap.add_argument("--msgs", type=int, default=14, help="log2 of the large hash batch")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU-port column")
    ap.add_argument("--lib", default=os.path.join(ROOT, "anemoi-rust_amd", "lib", "libanemoi_mi355x.so"))
    args = ap.parse_args()
    lib = ctypes.CDLL(os.path.abspath(args.lib))
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    lib.anemoi_jive_compress_k_dev.argtypes = [ci, ci, ci, vp, vp, sz, vp]
    lib.anemoi_hash_bytes_dev.argtypes = [ci, ci, vp, sz, sz, vp, vp]
    lib.anemoi_to_montgomery_dev.argtypes = [ci, vp, vp, sz, vp]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    rng = np.random.default_rng(3)

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            fn()
            b.record(stream)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return sorted(ts)[len(ts) // 2] * 1e3  # microseconds

    rows = []
    for name in args.fields.split(","):
        fid, L = FIELDS.index(name), LIMBS[FIELDS.index(name)]
        nbig = 1 << args.big
        one = np.zeros((nbig * 4, L), dtype=np.uint64)
        one[:, 0] = 1                                         # canonical 1 -> Felt::one() on the device
        d_one = torch.from_numpy(one.view(np.int64).reshape(-1)).to(dev)
        assert lib.anemoi_to_montgomery_dev(fid, d_one.data_ptr(), d_one.data_ptr(), nbig * 4, stream.cuda_stream) == 0
        d_out = torch.empty(nbig * 2 * L, dtype=torch.int64, device=dev)
        for label, width, k in (("compress/2-1", 2, 2), ("compress/4-3", 4, 2), ("compress_k4/4-3", 4, 4)):
            def run(n):
                assert lib.anemoi_jive_compress_k_dev(fid, width, k, d_one.data_ptr(), d_out.data_ptr(), n,
                                                      stream.cuda_stream) == 0
            single = timed(lambda: run(1))
            big = timed(lambda: run(nbig)) / nbig
            rows.append((name, label, single, big, README_US.get((name, label))))
        nm = 1 << args.msgs
        msgs = torch.from_numpy(rng.integers(0, 256, size=(nm, 10240), dtype=np.uint8)).to(dev)
        dig = torch.empty(nm * L, dtype=torch.int64, device=dev)
        for label, width in (("hash10KB/2-1", 2), ("hash10KB/4-3", 4)):
            def run(n):
                assert lib.anemoi_hash_bytes_dev(fid, width, msgs.data_ptr(), 10240, n, dig.data_ptr(),
                                                 stream.cuda_stream) == 0
            single = timed(lambda: run(1), reps=2)
            big = timed(lambda: run(nm), reps=2) / nm
            rows.append((name, label, single, big, README_US.get((name, label))))
    # the CPU port on THIS host, one call on one thread (what criterion measures), after all GPU timing
    cpu = {}
    if not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import cpu_port
        port = cpu_port.Port()
        print("CPU baseline: " + port.describe())
        for name in args.fields.split(","):
            fid, L = FIELDS.index(name), LIMBS[FIELDS.index(name)]
            for label, width, k in (("compress/2-1", 2, 2), ("compress/4-3", 4, 2), ("compress_k4/4-3", 4, 4)):
                cpu[(name, label)] = port.compress_us(fid, width, L, k=k, budget_s=0.6)[0]
            for label, width in (("hash10KB/2-1", 2), ("hash10KB/4-3", 4)):
                cpu[(name, label)] = port.hash_us(fid, width, 10240, budget_s=0.3)[0]
    print("%-16s %-16s %14s %18s %18s %16s" % ("field", "workload", "1 call [us]", "per item, batched", "CPU port, 1 call", "reference README"))
    for name, label, single, big, ref in rows:
        c = cpu.get((name, label))
        print("%-16s %-16s %14.1f %15.3f us %15s %16s" % (name, label, single, big, ("%.2f us" % c) if c else "-",
                                                           ("%.2f us" % ref) if ref else "-"))
    print(json.dumps([{"field": r[0], "workload": r[1], "single_call_us": r[2], "batched_us_per_item": r[3],
                       "cpu_port_us_one_thread": cpu.get((r[0], r[1])), "reference_readme_cpu_us": r[4]} for r in rows]))


if __name__ == "__main__":
    main()
