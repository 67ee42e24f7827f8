"""Range sharding for the one-process-per-GPU launch (bench.py, torch.distributed).

Batches of compressions / messages / Merkle subtrees are independent, so multi-GPU execution is a
contiguous range partition with NO data-path collective; the only cross-rank
traffic is the timing barrier / max-reduce and, for a Merkle tree, gathering one subtree root per rank.
"""


def shard_range(n, rank, world):
    """Items [begin, end) of rank `rank` out of `world`: contiguous, balanced to +-1, covering [0, n)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    return n * rank // world, n * (rank + 1) // world


def merkle_subtree_plan(depth, world):
    """(sub_log, sub_depth): 2^sub_log subtrees of depth sub_depth, one per rank (ranks >= 2^sub_log idle);
    the top sub_log levels are finished by rank 0 from the gathered subtree roots."""
    sub_log = 0
    while (2 << sub_log) <= world and sub_log + 1 <= depth:
        sub_log += 1
    return sub_log, depth - sub_log


def max_over_ranks(value, dist=None, device=None):
    """max of a float over all ranks (the bench's elapsed time); identity without a process group."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
