"""N > 1 path on CPU: two processes over gloo (127.0.0.1) shard a batch by contiguous ranges, each
processes its shard (with the oracle standing in for the GPU kernel -- this test checks the host
logic: partition, gather, max-over-ranks), and the gathered result must equal the unsharded run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def worker(rank, world, port, n, out_path):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "anemoi-rust_amd")):
        sys.path.insert(0, p)
    os.environ["ANEMOI_NO_TORCH_PRELOAD"] = "1"
    from anemoi_amd.shard import max_over_ranks, merkle_subtree_plan, shard_range
    import orc
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    oracle = orc.Oracle()
    fid = 6  # vesta
    rng = np.random.default_rng(42)
    states = rng.integers(0, 1 << 60, size=(n, 2, 4), dtype=np.uint64)  # same on every rank (same seed)
    b, e = shard_range(n, rank, world)
    mine = oracle.compress_batch(fid, 2, states[b:e], threads=1).reshape(-1, 4)
    gathered = [None] * world
    dist.all_gather_object(gathered, (b, e, mine))
    elapsed = max_over_ranks(0.5 + rank, dist)
    # Merkle: one subtree per rank, roots gathered, top finished by rank 0
    depth = 4
    leaves = oracle.compress_batch(fid, 2, states[: 2 << depth], threads=1).reshape(-1, 4)[: 1 << depth]
    sub_log, sub_depth = merkle_subtree_plan(depth, world)
    roots = [None] * world
    if rank < (1 << sub_log):
        sub = leaves[rank << sub_depth: (rank + 1) << sub_depth]
        my_root = oracle.merkle_root(fid, sub, sub_depth)
    else:
        my_root = None
    dist.all_gather_object(roots, my_root)
    if rank == 0:
        full = np.concatenate([g[2] for g in sorted(gathered, key=lambda g: g[0])])
        cover = sorted((g[0], g[1]) for g in gathered)
        top = oracle.merkle_root(fid, np.stack(roots[: 1 << sub_log]), sub_log)
        np.savez(out_path, full=full, cover=np.array(cover), elapsed=elapsed, top=top,
                 expect_top=oracle.merkle_root(fid, leaves, depth),
                 expect=oracle.compress_batch(fid, 2, states, threads=1).reshape(-1, 4))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_gloo(tmp_path):
    world, n = 2, 37  # odd size: ranges differ by one
    out = str(tmp_path / "out.npz")
    mp.spawn(worker, args=(world, free_port(), n, out), nprocs=world, join=True)
    r = np.load(out)
    assert r["cover"].tolist() == [[0, 18], [18, 37]]
    assert (r["full"] == r["expect"]).all()
    assert float(r["elapsed"]) == 1.5            # max over ranks of (0.5, 1.5)
    assert (r["top"] == r["expect_top"]).all()


def test_shard_range_properties():
    sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
    from anemoi_amd.shard import merkle_subtree_plan, shard_range
    for n in (0, 1, 7, 64, 1 << 20):
        for world in (1, 2, 3, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in edges]
            assert max(sizes) - min(sizes) <= 1
    assert merkle_subtree_plan(24, 8) == (3, 21)
    assert merkle_subtree_plan(24, 6) == (2, 22)
    assert merkle_subtree_plan(1, 8) == (1, 0)
    assert merkle_subtree_plan(0, 8) == (0, 0)
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)
