#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point (what the Rust shim calls) next to the
device-resident rate of the same kernel: anemoi_jive_compress_batch on BLS12-381 Anemoi-2-1 states in
pageable host memory, both staging modes (ANEMOI_HOST_STAGING=pinned|direct), 2^20 and 2^24 items, plus
the concurrent-callers overlap of latency-bound calls.

    python tools/bench_host_api.py [log2 sizes ...]      (default: 20 24)
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
import numpy as np
import torch
import anemoi_amd as A
from anemoi_amd import synth

sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or ([20, 24] if len(sys.argv) == 1 else [])
fid = A.field_id("bls12_381")
inst = A.Anemoi("bls12_381", 2)
inst.compress_batch(synth.states("bls12_381", 2, 1, 0, 4096))  # warm-up: constants, a lane


s = torch.cuda.current_stream()


def median(ts):
    return sorted(ts)[len(ts) // 2]


for lg in sizes:
    n = 1 << lg
    st = synth.states("bls12_381", 2, 0x5EED, 0, n)
    out = np.empty((n, 1, 6), dtype=np.uint64)
    # device-resident: inputs already in HBM, HIP events around the launch
    d_in = torch.from_numpy(st.view(np.int64).reshape(-1)).to("cuda:0")
    d_out = torch.empty(n * 6, dtype=torch.int64, device="cuda:0")
    s = torch.cuda.current_stream()
    res = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        assert A.lib.anemoi_jive_compress_k_dev(fid, 2, 2, d_in.data_ptr(), d_out.data_ptr(), n, s.cuda_stream) == 0
        b.record(s)
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b))
    resident = median(res[1:])
    want = d_out.cpu().numpy().view(np.uint64).reshape(n, 1, 6)
    del d_in, d_out
    torch.cuda.empty_cache()
    print("2^%d items: device-resident kernel %.2f ms = %.2f M/s" % (lg, resident, n / resident / 1e3))
    for mode in ("pinned", "direct"):
        A.set_option("host_staging", mode)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            rc = A.lib.anemoi_jive_compress_batch(fid, 2, st.ctypes.data_as(A._lib._u64p), out.ctypes.data_as(A._lib._u64p), n, 0)
            ts.append(time.perf_counter() - t0)
            assert rc == 0
        assert (out == want).all()
        t = median(ts[1:]) * 1e3
        print("  host-pointer, staging=%-6s: %.2f ms = %.2f M/s  -> %.3f x the resident rate (first call %.1f ms)"
              % (mode, t, n / t / 1e3, resident / t, ts[0] * 1e3))
    A.set_option("host_staging", None)

# config 3 through the host-pointer entry point: 2^16 messages x 10 240 bytes (640 MiB) of pageable memory.  Too few
# messages to cut into message chunks, so the library feeds them segment by segment (capi.hip sponge_segments).
if "cfg3" in sys.argv or len(sys.argv) == 1:
    cfg = synth.CFG3
    f3 = A.field_id(cfg["field"])
    msgs = synth.messages(cfg["seed"], 0, cfg["n"], cfg["msg_len"])
    dig = np.empty((cfg["n"], 4), dtype=np.uint64)
    d_m = torch.from_numpy(msgs).to("cuda:0")
    d_o = torch.empty(cfg["n"] * 4, dtype=torch.int64, device="cuda:0")
    res = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        assert A.lib.anemoi_hash_bytes_dev(f3, 4, d_m.data_ptr(), cfg["msg_len"], cfg["n"], d_o.data_ptr(), s.cuda_stream) == 0
        b.record(s)
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b))
    resident = median(res[1:])
    want = d_o.cpu().numpy().view(np.uint64).reshape(-1, 4)
    del d_m, d_o
    print("config 3 (2^16 x 10 240 B, BN-254 4-3): device-resident kernel %.1f ms" % resident)
    for label, env in (("segments (default)", None), ("single launch", "1099511627776")):
        if env:
            A.set_option("sponge_segment_bytes", int(env))   # one segment would hold everything -> single launch
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            rc = A.lib.anemoi_hash_bytes_batch(f3, 4, msgs.ctypes.data_as(A._lib._u8p), cfg["msg_len"], cfg["n"],
                                               dig.ctypes.data_as(A._lib._u64p), 0)
            ts.append(time.perf_counter() - t0)
            assert rc == 0
        assert (dig == want).all()
        t = median(ts[1:]) * 1e3
        print("  host-pointer, %-18s: %.1f ms -> %.3f x the resident rate (first call %.1f ms)" % (label, t, resident / t, ts[0] * 1e3))
    A.set_option("sponge_segment_bytes", None)

# A 640 MiB ragged batch (2^19 BN-254 4-3 messages of 1 024 .. 1 536 bytes, unsorted) through
# anemoi_hash_bytes_ragged_batch, which now runs on the chunked pipeline (chunks cut at message boundaries, offsets
# rebased per chunk, three chunks on the device) next to the same launch on a blob that is already in HBM.
if "ragged" in sys.argv or len(sys.argv) == 1:
    rng = np.random.default_rng(1)
    nm = 1 << 19
    lens = rng.integers(1024, 1537, size=nm).astype(np.uint64)
    offs = np.zeros(nm + 1, dtype=np.uint64)
    np.cumsum(lens, out=offs[1:])
    blob = rng.integers(0, 256, size=int(offs[-1]), dtype=np.uint8)
    f3 = A.field_id("bn_254")
    dig = np.empty((nm, 4), dtype=np.uint64)
    d_b, d_off = torch.from_numpy(blob).to("cuda:0"), torch.from_numpy(offs.view(np.int64)).to("cuda:0")
    d_o = torch.empty(nm * 4, dtype=torch.int64, device="cuda:0")
    res = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        assert A.lib.anemoi_hash_bytes_ragged_dev(f3, 4, d_b.data_ptr(), d_off.data_ptr(), nm, d_o.data_ptr(), s.cuda_stream) == 0
        b.record(s)
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b))
    resident = median(res[1:])
    want = d_o.cpu().numpy().view(np.uint64).reshape(-1, 4)
    d_scr = torch.empty(A.lib.anemoi_ragged_scratch_bytes(nm), dtype=torch.uint8, device="cuda:0")
    res = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        assert A.lib.anemoi_hash_bytes_ragged_bucketed_dev(f3, 4, d_b.data_ptr(), d_b.numel(), d_off.data_ptr(), nm, d_o.data_ptr(), d_scr.data_ptr(),
                                                           d_scr.numel(), s.cuda_stream) == 0
        b.record(s)
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b))
    assert (d_o.cpu().numpy().view(np.uint64).reshape(-1, 4) == want).all()
    bucketed = median(res[1:])
    del d_b, d_off, d_o, d_scr
    torch.cuda.empty_cache()
    print("ragged batch (2^19 messages, %.0f MiB, BN-254 4-3): device-resident kernel, messages in the given (unsorted) order %.1f ms; "
          "bucketed on the device (anemoi_hash_bytes_ragged_bucketed_dev) %.1f ms" % (blob.size / 2**20, resident, bucketed))
    for mode in ("pinned", "direct"):
        A.set_option("host_staging", mode)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            rc = A.lib.anemoi_hash_bytes_ragged_batch(f3, 4, blob.ctypes.data_as(A._lib._u8p), offs.ctypes.data_as(A._lib._u64p),
                                                      nm, dig.ctypes.data_as(A._lib._u64p), 0)
            ts.append(time.perf_counter() - t0)
            assert rc == 0
        assert (dig == want).all()
        t = median(ts[1:]) * 1e3
        print("  host-pointer, staging=%-6s: %.1f ms -> %.3f x the resident rate (first call %.1f ms)" % (mode, t, resident / t, ts[0] * 1e3))
    A.set_option("host_staging", None)
    del blob, offs
    # The same volume with a LONG-TAILED length mix (most messages ~200 bytes, one in sixteen 4-24 KB), unsorted as a
    # caller would hand it over, against the same messages pre-sorted by length: the library buckets by block count
    # while staging (host::ragged_order), so the unsorted batch should cost about what the sorted one does.
    nm = 1 << 19
    lens = np.where(rng.integers(0, 16, size=nm) == 0, rng.integers(4096, 24576, size=nm), rng.integers(64, 320, size=nm)).astype(np.uint64)
    offs = np.zeros(nm + 1, dtype=np.uint64)
    np.cumsum(lens, out=offs[1:])
    blob = rng.integers(0, 256, size=int(offs[-1]), dtype=np.uint8)
    order = np.argsort(-lens.astype(np.int64), kind="stable")
    s_offs = np.zeros(nm + 1, dtype=np.uint64)
    np.cumsum(lens[order], out=s_offs[1:])
    s_blob = np.empty_like(blob)
    starts = offs[:-1]
    for k, m in enumerate(order):
        s_blob[int(s_offs[k]):int(s_offs[k + 1])] = blob[int(starts[m]):int(starts[m]) + int(lens[m])]
    out_u, out_s = np.empty((nm, 4), dtype=np.uint64), np.empty((nm, 4), dtype=np.uint64)
    res = {}
    for label, b_, o_, d_ in (("pre-sorted by length", s_blob, s_offs, out_s), ("unsorted (library buckets)", blob, offs, out_u)):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            rc = A.lib.anemoi_hash_bytes_ragged_batch(f3, 4, b_.ctypes.data_as(A._lib._u8p), o_.ctypes.data_as(A._lib._u64p),
                                                      nm, d_.ctypes.data_as(A._lib._u64p), 0)
            ts.append(time.perf_counter() - t0)
            assert rc == 0
        res[label] = median(ts[1:]) * 1e3
    assert (out_u[order] == out_s).all()
    print("long-tailed ragged batch (2^19 messages, %.0f MiB): pre-sorted %.1f ms, unsorted %.1f ms -> %.3f x"
          % (blob.size / 2**20, res["pre-sorted by length"], res["unsorted (library buckets)"],
             res["unsorted (library buckets)"] / res["pre-sorted by length"]))
    # the same two batches DEVICE-RESIDENT: in the given order (anemoi_hash_bytes_ragged_dev) and bucketed on the device
    # (anemoi_hash_bytes_ragged_bucketed_dev: counting sort by block count, digests scattered back)
    def dev_ms(fn):
        ts = []
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            assert fn() == 0
            b.record(s)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return median(ts[1:])
    d_o = torch.zeros(nm * 4, dtype=torch.int64, device="cuda:0")
    d_scr = torch.empty(A.lib.anemoi_ragged_scratch_bytes(nm), dtype=torch.uint8, device="cuda:0")
    dres = {}
    for label, b_, o_ in (("pre-sorted", s_blob, s_offs), ("unsorted", blob, offs)):
        d_b, d_f = torch.from_numpy(b_).to("cuda:0"), torch.from_numpy(o_.view(np.int64)).to("cuda:0")
        dres[label, "in order"] = dev_ms(lambda: A.lib.anemoi_hash_bytes_ragged_dev(f3, 4, d_b.data_ptr(), d_f.data_ptr(), nm, d_o.data_ptr(), s.cuda_stream))
        ref = d_o.clone()
        dres[label, "bucketed"] = dev_ms(lambda: A.lib.anemoi_hash_bytes_ragged_bucketed_dev(
            f3, 4, d_b.data_ptr(), d_b.numel(), d_f.data_ptr(), nm, d_o.data_ptr(), d_scr.data_ptr(), d_scr.numel(), s.cuda_stream))
        assert torch.equal(ref, d_o)
        del d_b, d_f
    print("  device-resident: pre-sorted, in order %.1f ms | unsorted, in order %.1f ms | unsorted, bucketed on the device %.1f ms "
          "(%.3f x the pre-sorted one) | pre-sorted, bucketed %.1f ms"
          % (dres["pre-sorted", "in order"], dres["unsorted", "in order"], dres["unsorted", "bucketed"],
             dres["unsorted", "bucketed"] / dres["pre-sorted", "in order"], dres["pre-sorted", "bucketed"]))
    del blob, offs, s_blob, s_offs, d_o, d_scr
    torch.cuda.empty_cache()

# 2^22 depth-24 authentication paths (Jubjub; 3.4 GB of host memory) through anemoi_merkle_verify_batch
if "verify" in sys.argv or len(sys.argv) == 1:
    nv, depth = 1 << 22, 24
    fj = A.field_id("jubjub")
    leaves = synth.elements("jubjub", 5, 0, nv).reshape(nv, 4)
    base = synth.elements("jubjub", 6, 0, 1 << 16).reshape(-1, 4)
    paths = np.ascontiguousarray(base[np.random.default_rng(2).integers(0, 1 << 16, size=nv * depth)]).reshape(nv, depth, 4)
    idx = np.random.default_rng(3).integers(0, 1 << depth, size=nv).astype(np.uint64)
    d_l, d_i = torch.from_numpy(leaves.view(np.int64)).to("cuda:0"), torch.from_numpy(idx.view(np.int64)).to("cuda:0")
    d_p = torch.from_numpy(paths.view(np.int64).reshape(-1)).to("cuda:0")
    d_r = torch.empty(nv * 4, dtype=torch.int64, device="cuda:0")
    res = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        assert A.lib.anemoi_merkle_climb_dev(fj, d_l.data_ptr(), d_i.data_ptr(), d_p.data_ptr(), depth, nv, d_r.data_ptr(), s.cuda_stream) == 0
        b.record(s)
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b))
    resident = median(res[1:])
    roots = d_r.cpu().numpy().view(np.uint64).reshape(-1, 4)
    del d_l, d_i, d_p, d_r
    torch.cuda.empty_cache()
    root = roots[12345].copy()            # the one item that verifies
    print("path verification (2^22 paths of depth 24, Jubjub): device-resident kernel %.1f ms" % resident)
    ok = np.zeros(nv, dtype=np.uint8)
    for mode in ("pinned", "direct"):
        A.set_option("host_staging", mode)
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            rc = A.lib.anemoi_merkle_verify_batch(fj, leaves.ctypes.data_as(A._lib._u64p), idx.ctypes.data_as(A._lib._u64p),
                                                  paths.ctypes.data_as(A._lib._u64p), depth, nv, root.ctypes.data_as(A._lib._u64p),
                                                  ok.ctypes.data_as(A._lib._u8p), 0)
            ts.append(time.perf_counter() - t0)
            assert rc == 0
        assert ok[12345] == 1 and ok.sum() == (roots == root).all(axis=1).sum()
        t = min(ts) * 1e3
        print("  host-pointer, staging=%-6s: %.1f ms -> %.3f x the resident rate (first call %.1f ms)" % (mode, t, resident / t, ts[0] * 1e3))
    A.set_option("host_staging", None)
    del leaves, paths, idx

# concurrent callers: latency-bound calls (48 items = one wave-cooperative launch each) from 4 threads
sts = [synth.states("bls12_381", 2, 77 + k, 0, 48) for k in range(4)]
for s_ in sts:
    inst.compress_batch(s_)
reps = 8


def run(k):
    for _ in range(reps):
        inst.compress_batch(sts[k])


t0 = time.perf_counter()
for k in range(4):
    run(k)
serial = time.perf_counter() - t0
ths = [threading.Thread(target=run, args=(k,)) for k in range(4)]
t0 = time.perf_counter()
for th in ths:
    th.start()
for th in ths:
    th.join()
conc = time.perf_counter() - t0
print("4 threads x %d calls of 48 items: serial %.1f ms, concurrent %.1f ms (%.2f x)" % (reps, serial * 1e3, conc * 1e3, serial / conc))
