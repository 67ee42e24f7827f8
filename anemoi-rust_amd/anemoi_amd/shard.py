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


class ControlPlaneAudit:
    """Counts what crosses torch.distributed while it is installed: every tensor-carrying collective / point-to-point
    function of the module is wrapped, so that bench.py's line can state -- and a test can assert -- that nothing but
    8-byte scalars (the max-over-ranks of the elapsed time, the failure flag, the probe's minimum) and barriers ever
    crossed the control plane: there is no collective on the data path."""

    TENSOR_CALLS = ("all_reduce", "broadcast", "reduce", "all_gather", "all_gather_into_tensor", "gather", "scatter",
                    "reduce_scatter", "reduce_scatter_tensor", "all_to_all", "all_to_all_single", "send", "recv",
                    "isend", "irecv", "all_gather_object", "broadcast_object_list", "gather_object", "scatter_object_list")

    def __init__(self, dist):
        self.dist, self.calls, self.max_tensor_bytes, self._saved = dist, {}, 0, {}

    def _note(self, name, args, kwargs):
        import torch
        self.calls[name] = self.calls.get(name, 0) + 1
        if "object" in name:
            self.max_tensor_bytes = max(self.max_tensor_bytes, 1 << 62)   # pickled objects: never on this control plane
        stack = list(args) + list(kwargs.values())
        while stack:
            a = stack.pop()
            if isinstance(a, torch.Tensor):
                self.max_tensor_bytes = max(self.max_tensor_bytes, a.numel() * a.element_size())
            elif isinstance(a, (list, tuple)):
                stack.extend(a)

    def install(self):
        for name in self.TENSOR_CALLS + ("barrier",):
            fn = getattr(self.dist, name, None)
            if fn is None:
                continue
            self._saved[name] = fn

            def wrapped(*args, _fn=fn, _name=name, **kwargs):
                self._note(_name, args, kwargs)
                return _fn(*args, **kwargs)
            setattr(self.dist, name, wrapped)
        return self

    def remove(self):
        for name, fn in self._saved.items():
            setattr(self.dist, name, fn)
        self._saved = {}

    def report(self):
        return {"calls": dict(sorted(self.calls.items())), "max_tensor_bytes": self.max_tensor_bytes}
