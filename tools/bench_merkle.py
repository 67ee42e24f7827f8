#!/usr/bin/env python3
"""Merkle root timing for a field/depth on one GPU.

  device-resident leaves, one stream, level by level   (anemoi_merkle_root_dev)
  host leaves through anemoi_merkle_root with ANEMOI_VIRTUAL_DEVICES = 1, 2, 4, 8, 16: that many subtrees
  built concurrently, each on its own lane (own streams), top levels afterwards -- shows what running
  independent subtrees side by side buys on the latency-bound narrow levels.

    python tools/bench_merkle.py <field name> <depth> [per-level]
"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
import numpy as np, torch
import anemoi_amd as A
from anemoi_amd import synth

field, depth = sys.argv[1], int(sys.argv[2])
fid, L = A.field_id(field), synth.limbs_of(field)
leaves = synth.elements(field, 0xBEEF, 0, 1 << depth)
dev = torch.device("cuda", 0)
d_leaves = torch.from_numpy(leaves.view(np.int64).reshape(-1)).to(dev)
scratch = torch.empty((1 << depth) * L, dtype=torch.int64, device=dev)
d_root = torch.empty(L, dtype=torch.int64, device=dev)
s = torch.cuda.current_stream()


def run_dev():
    assert A.lib.anemoi_merkle_root_dev(fid, d_leaves.data_ptr(), depth, scratch.data_ptr(), d_root.data_ptr(), s.cuda_stream) == 0


run_dev(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s); run_dev(); b.record(s); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
root_dev = d_root.cpu().numpy().view(np.uint64)
print("%s depth %d, device-resident, one stream (coop2d_max=%s): %.2f ms" % (field, depth, A.get_option("coop2d_max"), sorted(ts)[1]))

if len(sys.argv) > 3:  # per-level times: one launch per level, events around each
    pc = []
    src, n = d_leaves, 1 << depth
    for l in range(depth):
        n //= 2
        dst = torch.empty(max(n, 1) * L, dtype=torch.int64, device=dev)
        A.lib.anemoi_jive_compress_k_dev(fid, 2, 2, src.data_ptr(), dst.data_ptr(), n, s.cuda_stream)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); A.lib.anemoi_jive_compress_k_dev(fid, 2, 2, src.data_ptr(), dst.data_ptr(), n, s.cuda_stream); b.record(s)
        torch.cuda.synchronize()
        pc.append((n, a.elapsed_time(b)))
        src = dst
    print("  per level (nodes: ms): " + "  ".join("%d: %.2f" % x for x in pc))
    print("  sum of levels %.2f ms" % sum(t for _, t in pc))

inst = A.Anemoi(field, 2, device=A.ALL_DEVICES)
for parts in (1, 2, 4, 8, 16, 32):
    A.set_option("virtual_devices", parts)
    inst.merkle_root(leaves, depth)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = inst.merkle_root(leaves, depth); ts.append(time.perf_counter() - t0)
    assert (r == root_dev).all()
    print("  host leaves, %2d concurrent subtrees: %.2f ms" % (parts, sorted(ts)[1] * 1e3))

# the retained-level builder from host leaves (every level copied back to the caller's tree array)
one = A.Anemoi(field, 2, device=0)
A.set_option("virtual_devices", None)
one.merkle_tree(leaves, depth)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); lv = one.merkle_tree(leaves, depth); ts.append(time.perf_counter() - t0)
assert (lv[-1][0] == root_dev).all()
print("  host leaves -> retained tree (all %d levels copied out, %d MiB): %.2f ms" % (depth + 1, ((2 << depth) - 1) * L * 8 >> 20, sorted(ts)[1] * 1e3))
