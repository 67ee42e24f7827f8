#!/usr/bin/env python3
"""Times every BASELINE.json config shape on ONE GPU (inputs resident in HBM, HIP events on the launch
stream).  Not the driver's bench (that is bench.py = config 2); this fills the per-row measurements
of DESIGN.md / profiles/README.md.
    python tools/bench_configs.py [--lib path.so] [--quick]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd", "anemoi_amd"))
import numpy as np
import torch
import buildinfo  # noqa: E402  (multiply-add counts per compression, from the generated assembly and schedules)

# peak v_mad_u64_u32 lane-operations per second: 1024 SIMDs x 16 lanes per clock at the nominal 2.4 GHz (the boxes
# run at 2.30-2.36 GHz under this load: GRBM_GUI_ACTIVE in profiles/rNN/pmc_*.json), so alu_frac here is a lower bound
PEAK_LANE_MAD = 1024 * 16 * 2.4e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "anemoi-rust_amd", "lib", "libanemoi_mi355x.so"))
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU-port column")
    args = ap.parse_args()
    lib = ctypes.CDLL(os.path.abspath(args.lib))
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    lib.anemoi_jive_compress_k_dev.argtypes = [ci, ci, ci, vp, vp, sz, vp]
    lib.anemoi_hash_bytes_dev.argtypes = [ci, ci, vp, sz, sz, vp, vp]
    lib.anemoi_merkle_root_dev.argtypes = [ci, vp, ctypes.c_uint, vp, vp, vp]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    rng = np.random.default_rng(7)

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
        return sorted(ts)[len(ts) // 2]

    def states(n, width, limbs):
        h = rng.integers(0, 1 << 60, size=(n, width, limbs), dtype=np.uint64)  # limbs < 2^60 => element < p
        return torch.from_numpy(h.view(np.int64).reshape(-1)).to(dev)

    out = {}

    def jive(name, field, width, limbs, n, modmul):
        d_in, d_out = states(n, width, limbs), torch.empty(n * limbs * (width // 2), dtype=torch.int64, device=dev)
        ms = timed(lambda: check(lib.anemoi_jive_compress_k_dev(field, width, 2, d_in.data_ptr(), d_out.data_ptr(), n,
                                                                stream.cuda_stream)))
        mad = buildinfo.mad_per_compression(field, width)
        out[name] = {"ms": ms, "compress_per_s": n / ms * 1e3, "modmul_per_s": n * modmul / ms * 1e3,
                     "algorithmic_GBps": n * limbs * 8 * (width + width // 2) / ms / 1e6,
                     "alu_frac": n * mad / (ms / 1e3) / PEAK_LANE_MAD}

    def check(rc):
        assert rc == 0, rc

    jive("cfg1_vesta_2_1_x1024", 6, 2, 4, 1024, 6195)
    jive("cfg2_bls12_381_2_1_x2^20", 0, 2, 6, 1 << 20, 9576)
    if not args.quick:
        jive("cfg4_bls12_381_2_1_x2^21_per_gpu", 0, 2, 6, 1 << 21, 9576)
    jive("extra_jubjub_2_1_x2^20", 4, 2, 4, 1 << 20, 6447)
    jive("extra_bn254_4_3_x2^20", 2, 4, 4, 1 << 20, 8596)
    jive("extra_bls12_381_4_3_x2^19", 0, 4, 6, 1 << 19, 12768)

    # config 3: Anemoi-4-3 over BN-254, sponge hash of 10 240-byte messages
    nmsg = (1 << 13) if args.quick else (1 << 16)
    msgs = torch.from_numpy(rng.integers(0, 256, size=(nmsg, 10240), dtype=np.uint8)).to(dev)
    dig = torch.empty(nmsg * 4, dtype=torch.int64, device=dev)
    ms = timed(lambda: check(lib.anemoi_hash_bytes_dev(2, 4, msgs.data_ptr(), 10240, nmsg, dig.data_ptr(),
                                                       stream.cuda_stream)), reps=2)
    # per message: 111 permutations on a lane pair + 331 element decodes (one product each, on both lanes)
    mad3 = 111 * buildinfo.mad_per_permutation(2, 4) + 2 * 331 * buildinfo.limb_layout(2)["mul_mad"]
    flat43 = out["extra_bn254_4_3_x2^20"]["compress_per_s"]
    out["cfg3_bn254_4_3_sponge_10KB_x2^%d" % (13 if args.quick else 16)] = {
        "ms": ms, "messages_per_s": nmsg / ms * 1e3, "permutations_per_s": nmsg * 111 / ms * 1e3,
        "modmul_per_s": nmsg * 954156 / ms * 1e3, "algorithmic_GBps": nmsg * 10272 / ms / 1e6,
        "alu_frac": nmsg * mad3 / (ms / 1e3) / PEAK_LANE_MAD,
        # permutations per second against the flat BN-254 4-3 Jive rate of this run (one permutation per compression)
        "fraction_of_flat_rate": (nmsg * 111 / ms * 1e3) / flat43}

    # config 5: Jubjub Merkle tree, one GPU's subtree (depth 21 of the depth-24 tree; 2^21 leaves)
    depth = 17 if args.quick else 21
    leaves = states(1 << depth, 1, 4)
    scratch = torch.empty((1 << depth) * 4, dtype=torch.int64, device=dev)
    root = torch.empty(4, dtype=torch.int64, device=dev)
    ms = timed(lambda: check(lib.anemoi_merkle_root_dev(4, leaves.data_ptr(), depth, scratch.data_ptr(), root.data_ptr(),
                                                        stream.cuda_stream)), reps=2)
    merges = (1 << depth) - 1
    out["cfg5_jubjub_merkle_depth%d_subtree" % depth] = {
        "ms": ms, "merges_per_s": merges / ms * 1e3, "algorithmic_GBps": 96 * merges / ms / 1e6,
        "alu_frac": merges * buildinfo.mad_per_compression(4, 2) / (ms / 1e3) / PEAK_LANE_MAD,
        # merges per second against the flat Jubjub 2-1 Jive rate of this run: the tree's top levels are latency-bound
        "fraction_of_flat_rate": (merges / ms * 1e3) / out["extra_jubjub_2_1_x2^20"]["compress_per_s"]}
    # ---- the CPU port beside every config (after and outside all GPU-timed regions) ----------
    if not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import cpu_port
        port = cpu_port.Port()
        print("CPU baseline: " + port.describe())
        # config 1 whole (1 024 Vesta compressions: BASELINE.json defines it as a CPU configuration)
        one_us, rate = port.compress_us(6, 2, 4)
        out["cfg1_vesta_2_1_x1024"]["cpu_port"] = {"us_per_compress_1_thread": one_us, "compress_per_s_all_threads": rate,
                                                   "ms_for_1024_all_threads": 1024 / rate * 1e3, "threads": port.threads}
        for key, fid, width, limbs in (("cfg2_bls12_381_2_1_x2^20", 0, 2, 6), ("cfg4_bls12_381_2_1_x2^21_per_gpu", 0, 2, 6),
                                        ("extra_jubjub_2_1_x2^20", 4, 2, 4), ("extra_bn254_4_3_x2^20", 2, 4, 4),
                                        ("extra_bls12_381_4_3_x2^19", 0, 4, 6)):
            if key in out:
                one_us, rate = port.compress_us(fid, width, limbs)
                out[key]["cpu_port"] = {"us_per_compress_1_thread": one_us, "compress_per_s_all_threads": rate,
                                        "threads": port.threads, "gpu_over_cpu_all_threads": out[key]["compress_per_s"] / rate}
        # config 3 down-scaled: 2^6 messages' worth of time per thread, the full 10 240-byte message
        k3 = [k for k in out if k.startswith("cfg3")][0]
        one_us, rate = port.hash_us(2, 4, 10240, budget_s=3.0)
        out[k3]["cpu_port"] = {"ms_per_message_1_thread": one_us / 1e3, "messages_per_s_all_threads": rate, "threads": port.threads,
                               "gpu_over_cpu_all_threads": out[k3]["messages_per_s"] / rate}
        # config 5 down-scaled: a depth-14 Jubjub tree on all threads (the GPU's depth-21 subtree has 128 x the merges)
        k5 = [k for k in out if k.startswith("cfg5")][0]
        ms14 = port.merkle_ms(4, 4, 14)
        out[k5]["cpu_port"] = {"ms_depth14_all_threads": ms14, "merges_per_s_all_threads": ((1 << 14) - 1) / ms14 * 1e3,
                               "threads": port.threads,
                               "gpu_over_cpu_all_threads": out[k5]["merges_per_s"] / (((1 << 14) - 1) / ms14 * 1e3)}
    for k, v in out.items():
        print("%-44s %s" % (k, json.dumps({a: (b if isinstance(b, dict) else (round(b, 3) if b < 1e4 else float("%.4g" % b)))
                                            for a, b in v.items()})))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
