#!/usr/bin/env python3
"""Does any re-ordering of ONE tree's work hide its latency-bound tail?  DESIGN.md section 6 argues no (the last piece's tail
is as long as the whole tree's); this measures it.  A depth-21 Jubjub tree (config 5's share of one GPU) built
  (a) level by level in one call (anemoi_merkle_root_dev);
  (b) as 2 / 4 / 8 subtrees, each a call of its own on a stream of its own, all enqueued at once (the hardware interleaves
      them as it likes), the top levels on the first stream behind events;
  (c) the same with the subtrees' streams at descending priority (subtree 0 first wherever it has work, the others fill in).
Every variant's root is compared with (a)'s.  Times: hipEvents around each variant, best of 5.
    python tools/exp_merkle_staggered.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
import numpy as np
import torch
import anemoi_amd as A

dev = torch.device("cuda", 0)
jub, depth, L = 4, 21, 4
assert A.lib.anemoi_init(0, jub, 2) == 0
A.warmup("jubjub", 2)
rng = np.random.default_rng(1)
leaves = rng.integers(0, 1 << 60, size=(1 << depth, L), dtype=np.uint64)
d_leaves = torch.from_numpy(leaves.view(np.int64).reshape(-1)).to(dev)
d_scratch = torch.empty((1 << depth) * L, dtype=torch.int64, device=dev)
d_roots = torch.zeros(16 * L, dtype=torch.int64, device=dev)
d_top = torch.zeros(16 * L, dtype=torch.int64, device=dev)
d_root = torch.zeros(L, dtype=torch.int64, device=dev)
main = torch.cuda.Stream()


def whole(stream):
    assert A.lib.anemoi_merkle_root_dev(jub, d_leaves.data_ptr(), depth, d_scratch.data_ptr(), d_root.data_ptr(), stream.cuda_stream) == 0


def split(parts_log, streams):
    sub = depth - parts_log
    per = (1 << sub) * L * 8
    evs = []
    for i in range(1 << parts_log):
        s = streams[i]
        assert A.lib.anemoi_merkle_root_dev(jub, d_leaves.data_ptr() + i * per, sub, d_scratch.data_ptr() + i * per,
                                            d_roots.data_ptr() + i * L * 8, s.cuda_stream) == 0
        if i:
            e = torch.cuda.Event()
            e.record(s)
            evs.append(e)
    for e in evs:
        streams[0].wait_event(e)
    assert A.lib.anemoi_merkle_root_dev(jub, d_roots.data_ptr(), parts_log, d_top.data_ptr(), d_root.data_ptr(), streams[0].cuda_stream) == 0


def timed(fn, wait_on):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(wait_on)
        fn()
        b.record(wait_on)
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best, d_root.cpu().numpy().copy()


t, ref = timed(lambda: whole(main), main)
print("(a) one call, level by level:                        %7.2f ms" % t)
for parts_log in (1, 2, 3):
    plain = [torch.cuda.Stream() for _ in range(1 << parts_log)]
    t, r = timed(lambda: split(parts_log, plain), plain[0])
    assert (r == ref).all()
    print("(b) %d subtrees on %d streams, enqueued at once:       %7.2f ms" % (1 << parts_log, 1 << parts_log, t))
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    prio = [torch.cuda.Stream(priority=(-1 if i == 0 else 0)) for i in range(1 << parts_log)]
    t, r = timed(lambda: split(parts_log, prio), prio[0])
    assert (r == ref).all()
    print("(c) the same, subtree 0's stream at high priority:    %7.2f ms" % t)
