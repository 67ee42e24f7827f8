#!/usr/bin/env python3
"""Experiment: does a Merkle tree finish sooner when independent subtrees climb on separate HIP streams?

The narrow levels of a tree (<= 65 536 nodes) sit on a latency floor (one compression = 1.4 ms Jubjub / 3 ms
BLS12-381) while most of the chip idles.  Levels of DIFFERENT subtrees are independent, so in principle their
narrow levels can run under another subtree's wide levels.  This script builds the same device-resident tree

    mode "one"      level by level on one stream (what anemoi_merkle_root_dev does today)
    mode "chains"   S subtrees, each a complete chain of levels on its own stream (K streams round-robin)
    mode "split"    the wide levels (>= WIDE nodes) of the whole tree on one stream, then the rest as S subtrees on
                    their own streams

and prints the time of each and checks every root against mode "one".

    python tools/exp_merkle_streams.py <field> <depth> [wide=131072]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
import numpy as np, torch
import anemoi_amd as A
from anemoi_amd import synth

field, depth = sys.argv[1], int(sys.argv[2])
WIDE = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
fid, L = A.field_id(field), synth.limbs_of(field)
dev = torch.device("cuda", 0)
leaves = synth.elements(field, 0xBEEF, 0, 1 << depth)
d_leaves = torch.from_numpy(leaves.view(np.int64).reshape(-1)).to(dev)
# every level of the tree, back to back (level l has 2^(depth-l) nodes)
levels = [d_leaves] + [torch.empty((1 << (depth - l)) * L, dtype=torch.int64, device=dev) for l in range(1, depth + 1)]


def jive(src, dst, n, stream):
    rc = A.lib.anemoi_jive_compress_k_dev(fid, 2, 2, src, dst, n, stream.cuda_stream)
    assert rc == 0, rc


def ptr(level, node):
    return levels[level].data_ptr() + node * L * 8


def run_one(main):
    for l in range(1, depth + 1):
        jive(ptr(l - 1, 0), ptr(l, 0), 1 << (depth - l), main)


def climb(first_level, sub_log, main, streams):
    """levels first_level+1 .. depth: S = 2^sub_log subtrees each on a stream up to the level with S nodes, then the top"""
    S = 1 << sub_log
    top_start = depth - sub_log      # level with S nodes
    start = torch.cuda.Event()
    start.record(main)
    done = []
    for k in range(S):
        st = streams[k % len(streams)]
        st.wait_event(start)
        for l in range(first_level + 1, top_start + 1):
            n = 1 << (depth - l - sub_log)       # this subtree's nodes at level l
            jive(ptr(l - 1, 2 * n * k), ptr(l, n * k), n, st)
        e = torch.cuda.Event()
        e.record(st)
        done.append(e)
    for e in done:
        main.wait_event(e)
    for l in range(top_start + 1, depth + 1):
        jive(ptr(l - 1, 0), ptr(l, 0), 1 << (depth - l), main)


def run_chains(main, streams, sub_log):
    climb(0, sub_log, main, streams)


def run_split(main, streams, sub_log):
    l = 0
    while l < depth and (1 << (depth - l - 1)) >= WIDE:
        l += 1
        jive(ptr(l - 1, 0), ptr(l, 0), 1 << (depth - l), main)
    if depth - l < sub_log:
        sub_log = depth - l
    climb(l, sub_log, main, streams)


def timed(fn, *a):
    main = torch.cuda.current_stream()
    fn(main, *a)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        levels[depth].zero_()
        a0, b0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(main); fn(main, *a); b0.record(main)
        torch.cuda.synchronize()
        ts.append(a0.elapsed_time(b0))
    return sorted(ts)[1], levels[depth].cpu().numpy().copy()


print("GPU_MAX_HW_QUEUES=%s  %s depth %d  wide >= %d" % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), field, depth, WIDE), flush=True)
t1, root = timed(run_one)
print("one stream, level by level: %.2f ms" % t1, flush=True)
for prio in (False, True):
    for nstreams in (2, 4, 8):
        streams = [torch.cuda.Stream(priority=-1 if prio else 0) for _ in range(nstreams)]
        for sub_log in (1, 2, 3, 4, 5):
            S = 1 << sub_log
            if sub_log > depth - 2:
                continue
            t, r = timed(run_chains, streams, sub_log)
            assert (r == root).all()
            t2, r2 = timed(run_split, streams, sub_log)
            assert (r2 == root).all()
            print("  %d streams%s, %2d subtrees: chains %.2f ms   split(wide on main) %.2f ms" % (nstreams, " (high priority)" if prio else "", S, t, t2), flush=True)
