#!/usr/bin/env python3
"""A/B timing of several builds of libanemoi_mi355x.so in ONE process on ONE device, interleaved
rounds (cdna_hip_programming.md §5.4 rule 24: different boxes differ by >10 %, so never compare
numbers across gpurun calls).  Usage:
    python tools/ab_bench.py [--log2 20] [--rounds 5] name=path.so name=path.so ...
Prints median / min kernel ms and compressions/s for each build (BLS12-381 Anemoi-2-1 Jive)."""
import argparse
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd", "anemoi_amd"))
import torch  # noqa: E402  (first: one HIP runtime for everything)
import synth  # noqa: E402  (plain module import; the libraries under test are loaded explicitly below)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--field", type=int, default=0)
    ap.add_argument("--width", type=int, default=2)
    ap.add_argument("libs", nargs="+")
    args = ap.parse_args()
    n = 1 << args.log2
    limbs = 6 if args.field in (0, 1) else 4
    dev = torch.device("cuda", 0)
    fields = ["bls12_381", "bls12_377", "bn_254", "ed_on_bls12_377", "jubjub", "pallas", "vesta"]
    host = synth.states(fields[args.field], args.width, 1, 0, n)
    d_in = torch.from_numpy(host.view("int64").reshape(-1)).to(dev)
    d_out = torch.empty(n * limbs * (args.width // 2), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()
    libs = []
    for spec in args.libs:
        name, path = spec.split("=", 1)
        lib = ctypes.CDLL(os.path.abspath(path))
        fn = lib.anemoi_jive_compress_k_dev
        fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                       ctypes.c_void_p]
        fn.restype = ctypes.c_int
        libs.append((name, fn, []))
    ref = None
    for name, fn, _ in libs:  # warm-up + cross-check outputs between builds
        assert fn(args.field, args.width, 2, d_in.data_ptr(), d_out.data_ptr(), n, stream.cuda_stream) == 0
        torch.cuda.synchronize()
        if ref is None:
            ref = d_out.clone()
        else:
            assert torch.equal(ref, d_out), "build %s disagrees with %s" % (name, libs[0][0])
    for _ in range(args.rounds):
        for name, fn, times in libs:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            assert fn(args.field, args.width, 2, d_in.data_ptr(), d_out.data_ptr(), n, stream.cuda_stream) == 0
            b.record(stream)
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b))
    for name, _, times in libs:
        med = statistics.median(times)
        print("%-24s median %8.2f ms  min %8.2f ms  -> %6.2f M compress/s (median)" % (name, med, min(times), n / med / 1e3))


if __name__ == "__main__":
    main()
