"""bench.py's N > 1 host logic on CPU, 2 / 4 / 8 processes over gloo -- 8 is the driver's first multi-GPU launch, which no
box available to the builder can run: rank -> shard of config 4, the seeded generator producing a rank's slice on its
own, the comparison of a rank's outputs with the committed oracle goldens (here the oracle stands in for the kernel on
the sampled items only -- 2^21 compressions per rank are a GPU-sized job), the per-shard SHA-256 selection, the
out-of-range guard, and the all-ranks failure flag that makes every rank exit when one rank's -- the LAST rank's --
output is wrong."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def worker(rank, world, port, corrupt_rank, out_path):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "anemoi-rust_amd")):
        sys.path.insert(0, p)
    os.environ["ANEMOI_NO_TORCH_PRELOAD"] = "1"
    import bench
    import orc
    from anemoi_amd import synth
    from anemoi_amd.shard import max_over_ranks
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    cfg_name, cfg, first, n = bench.rank_shard(rank, world)
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg_full.json")))[cfg_name]
    oracle = orc.Oracle()
    stride = golden["sample_stride"]
    idx = np.arange(((first + stride - 1) // stride) * stride, first + n, stride)
    # the rank's slice of the generator, item by item, must equal the same items of a bigger slice
    st = np.concatenate([synth.states(cfg["field"], 2, cfg["seed"], int(i), 1) for i in idx])
    assert (st[:8] == synth.states(cfg["field"], 2, cfg["seed"], int(idx[0]), 8 * stride)[::stride]).all()
    out = np.zeros((n, 6), dtype=np.uint64)
    out[idx - first] = oracle.compress_batch(0, 2, st, threads=2).reshape(-1, 6)
    if rank == corrupt_rank:
        out[idx[3] - first, 2] ^= np.uint64(1)
    checked, sha_ok, err = bench.verify_against_golden(out, golden, first, n, check_sha=False)
    any_failed = max_over_ranks(1.0 if err else 0.0, dist) != 0.0
    res = [None] * world
    dist.all_gather_object(res, (cfg_name, first, n, checked, err, any_failed))
    if rank == 0:
        json.dump(res, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


def run(tmp_path, world, corrupt_rank):
    out = str(tmp_path / ("res_%d_%d.json" % (world, corrupt_rank)))
    mp.spawn(worker, args=(world, free_port(), corrupt_rank, out), nprocs=world, join=True)
    return json.load(open(out))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_every_rank_verifies_its_config4_shard(tmp_path, world):
    res = run(tmp_path, world, -1)
    assert [tuple(r[:3]) for r in res] == [("cfg4", r << 21, 1 << 21) for r in range(world)]
    assert all(r[3] == 512 and r[4] is None and r[5] is False for r in res)


@pytest.mark.parametrize("world", [2, 8])
def test_a_wrong_last_rank_fails_every_rank(tmp_path, world):
    res = run(tmp_path, world, world - 1)
    assert all(r[4] is None for r in res[:-1]) and res[-1][4] and "item" in res[-1][4]
    assert all(r[5] is True for r in res)      # every rank knows, every rank would exit non-zero


def test_shard_digest_selection_and_the_out_of_range_guard():
    """verify_against_golden picks shard_sha256[first // n]: every one of 8 ranks must find ITS digest (a fake golden of
    the real one's shape: 8 shards, a strided sample), a rank given another rank's outputs must not; rank_shard refuses
    a rank whose shard would reach beyond the config's batch (9 ranks of 2^21, or 8 ranks of 2^22)."""
    import hashlib
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
    os.environ["ANEMOI_NO_TORCH_PRELOAD"] = "1"
    import bench
    per, shards, stride = 64, 8, 16
    rng = np.random.default_rng(9)
    outs = rng.integers(0, 1 << 63, size=(per * shards, 6), dtype=np.uint64)
    golden = {"n": per * shards, "shards": shards, "sample_stride": stride,
              "sample": [outs[i].tobytes().hex() for i in range(0, per * shards, stride)],
              "sha256": hashlib.sha256(outs.tobytes()).hexdigest(),
              "shard_sha256": [hashlib.sha256(outs[r * per:(r + 1) * per].tobytes()).hexdigest() for r in range(shards)]}
    for r in range(shards):
        checked, sha_ok, err = bench.verify_against_golden(outs[r * per:(r + 1) * per], golden, r * per, per)
        assert (checked, sha_ok, err) == (per // stride, True, None), r
        other = outs[((r + 1) % shards) * per:((r + 1) % shards + 1) * per]
        assert bench.verify_against_golden(other, golden, r * per, per)[2] is not None, r
        bent = outs[r * per:(r + 1) * per].copy()
        bent[stride + 1, 0] ^= np.uint64(1)                   # not a sampled item: only the shard's SHA-256 sees it
        assert "SHA-256" in bench.verify_against_golden(bent, golden, r * per, per)[2], r
    assert bench.verify_against_golden(outs, golden, 0, per * shards)[:2] == (per * shards // stride, True)
    with pytest.raises(SystemExit):
        bench.rank_shard(8, 9)
    with pytest.raises(SystemExit):
        bench.rank_shard(4, 8, batch_log2=22)
    assert bench.rank_shard(7, 8)[2:] == (7 << 21, 1 << 21)


def test_single_gpu_shard_is_config2():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
    os.environ["ANEMOI_NO_TORCH_PRELOAD"] = "1"
    import bench
    name, cfg, first, n = bench.rank_shard(0, 1)
    assert (name, first, n, cfg["seed"]) == ("cfg2", 0, 1 << 20, 0xA9E30102)
    name, cfg, first, n = bench.rank_shard(7, 8)
    assert (name, first, n, first + n) == ("cfg4", 7 << 21, 1 << 21, 1 << 24)
    # ANEMOI_BENCH_FORCE_DIST=1: one rank takes the N > 1 path = shard 0 of config 4
    name, cfg, first, n = bench.rank_shard(0, 1, sharded=True)
    assert (name, first, n, cfg["seed"]) == ("cfg4", 0, 1 << 21, 0xA9E30104)


def audit_worker(rank, world, port, out_path):
    sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
    os.environ["ANEMOI_NO_TORCH_PRELOAD"] = "1"
    import torch
    from anemoi_amd.shard import ControlPlaneAudit, max_over_ranks
    audit = ControlPlaneAudit(dist).install()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    dist.barrier()
    assert max_over_ranks(float(rank), dist) == world - 1
    clean = audit.report()
    dist.all_reduce(torch.zeros(1024, dtype=torch.float32))          # what a data-path collective would look like
    big = audit.report()
    audit.remove()
    dist.all_reduce(torch.zeros(4096, dtype=torch.float32))          # no longer counted
    after = audit.report()
    if rank == 0:
        json.dump([clean, big, after], open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_control_plane_audit_counts_every_tensor(tmp_path):
    """bench.py's line states what crossed torch.distributed (config.control_plane_traffic); the wrapper that counts it sees
    the barrier, the 8-byte scalar of max_over_ranks, and -- were one ever added -- a data-path collective."""
    out = str(tmp_path / "audit.json")
    mp.spawn(audit_worker, args=(2, free_port(), out), nprocs=2, join=True)
    clean, big, after = json.load(open(out))
    assert clean == {"calls": {"all_reduce": 1, "barrier": 1}, "max_tensor_bytes": 8}
    assert big == {"calls": {"all_reduce": 2, "barrier": 1}, "max_tensor_bytes": 4096}
    assert after == big
