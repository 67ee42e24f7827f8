#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: Jive 2-to-1 compressions/s (Anemoi-2-1, BLS12-381).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (anemoi_jive_compress_k_dev, k = 2) over one batch of 2^20
synthetic states that are already resident in HBM (configs[1] of BASELINE.json).  With N > 1 every
rank processes its own 2^20-state batch on its own GPU (weak scaling; items are independent, so
there is no data-path collective -- SURVEY.md §8e); `value` = all ranks' items / max-over-ranks time.

PyTorch is plumbing only (device buffers, the stream, torch.distributed for the barrier/max); the
work is done by libanemoi_mi355x.so through its C-ABI.

Also printed in the same JSON line:
  roofline      algorithmic bytes (144 B per compression: 96 in + 48 out) / the kernel's average
                launch duration measured with HIP events on the launch stream, against 8 TB/s HBM.
                The path is VALU-bound by ~4 orders of magnitude (9 576 384-bit modmul per 144 B), so
                this fraction is tiny by construction; `alu` carries the meaningful efficiency figure.
  cpu_baseline  the pinned C oracle ("port": same algorithm, u64-limb CIOS like arkworks) timed on a
                bounded sample on this box's host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "anemoi-rust_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

FIELD, WIDTH, LIMBS = "bls12_381", 2, 6
BATCH_LOG2 = 20
BYTES_PER_ITEM = 96 + 48          # 2 x 48 B in + 48 B out
MODMUL_PER_ITEM = 9576            # 21 rounds x (454 + 2), reference chain
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E peak
P_BLS12_381 = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def synth_states(n, seed):
    """n x [2][6] u64 limbs, every element < p (any value < p is a valid Montgomery element)."""
    rng = np.random.default_rng(seed)
    st = rng.integers(0, 1 << 64, size=(n, WIDTH, LIMBS), dtype=np.uint64)
    top = P_BLS12_381 >> 320
    st[:, :, LIMBS - 1] = rng.integers(0, top, size=(n, WIDTH), dtype=np.uint64)  # top limb < p's top limb
    return st


def usable_cores():
    """Host threads this process may really use: the CPU affinity capped by the cgroup quota."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    # a 1-GPU box grants this job a 16-CPU share of the host whatever the affinity mask says
    return max(1, min(cores, int(os.environ.get("ANEMOI_CPU_THREADS", "16"))))


def cpu_baseline(budget_s=12.0):
    """Oracle timed on a bounded sample of the same workload (rank 0, N = 1)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    path = None
    try:  # prefer a -march=native build for this host
        path = orc.build(native=True, out=os.path.join("/tmp", "liboracle_native_%d.so" % os.getpid()))
    except Exception:
        path = None
    oracle = orc.Oracle(path)
    cores = usable_cores()
    st = synth_states(64 * cores, 0xC0)
    t0 = time.perf_counter()
    oracle.compress_batch(0, WIDTH, st, threads=cores)
    rate = len(st) / (time.perf_counter() - t0)
    n = int(max(64 * cores, min(rate * budget_s, 1 << 20)))
    st = synth_states(n, 0xC1)
    t0 = time.perf_counter()
    oracle.compress_batch(0, WIDTH, st, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "compressions/s", "cores": cores, "kind": "port",
            "sample": "%d BLS12-381 Anemoi-2-1 Jive compressions, C oracle (u64 CIOS Montgomery), %d threads, %.1f s"
                      % (n, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-log2", type=int, default=BATCH_LOG2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    # one process per GPU; ANEMOI_BENCH_BACKEND=gloo lets the N > 1 path be rehearsed on a box with
    # fewer GPUs than ranks (ranks then share devices round-robin; timing is meaningless there)
    backend = os.environ.get("ANEMOI_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    local_rank = local_rank % max(ndev, 1) if backend == "gloo" else local_rank
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            try:  # create the RCCL communicator now, outside the timed region
                dist.barrier()
            except Exception as e:  # the data path has no collective: the control plane may fall back to gloo
                sys.stderr.write("rank %d: RCCL barrier failed (%s); control plane falls back to gloo\n" % (rank, e))
                dist.destroy_process_group()
                backend = "gloo"
                dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import anemoi_amd as A  # after torch: binds to the same HIP runtime
    from anemoi_amd.shard import max_over_ranks
    fid = A.field_id(FIELD)
    n = 1 << args.batch_log2
    dev = torch.device("cuda", local_rank)
    host = synth_states(n, 0xA9E30102 + rank)
    d_in = torch.from_numpy(host.view(np.int64).reshape(-1)).to(dev)
    d_out = torch.empty(n * LIMBS, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()

    def step():
        rc = A.lib.anemoi_jive_compress_k_dev(fid, WIDTH, 2, d_in.data_ptr(), d_out.data_ptr(), n, stream.cuda_stream)
        if rc != 0:
            raise A.AnemoiError(rc, A.lib.anemoi_last_error().decode())

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(max(args.warmup, 0)):
        step()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed = max_over_ranks(elapsed, dist, dev if backend == "nccl" else None)
    kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / max(len(evs), 1)

    # spot-check the timed output against the C-ABI's own single-item path is the tests' job; here only
    # make sure the kernel wrote something other than the input pattern
    assert int(torch.count_nonzero(d_out[: 6 * 64]).item()) > 0

    if rank == 0:
        # HBM traffic and VALU utilisation come from the committed rocprofv3 --pmc passes of this same
        # command (profiles/rNN/pmc_k_jive.json): counters cannot be read from inside the process.
        traffic, valu_busy, mad_frac, prof_src = None, None, None, None
        try:
            prof_dirs = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.startswith("r"))
            prof_src = os.path.join("profiles", prof_dirs[-1], "pmc_k_jive.json")
            pmc = json.load(open(os.path.join(ROOT, prof_src)))["derived"]
            if args.batch_log2 == BATCH_LOG2:
                traffic = pmc["hbm_traffic_bytes_per_launch"]
            valu_busy = pmc["valu_issue_model_fraction"]
            mad_frac = pmc["mad_cycle_fraction"]
        except Exception:
            prof_src = None
        total_items = n * world * args.steps
        value = total_items / elapsed
        achieved = BYTES_PER_ITEM * n / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "Jive 2-to-1 compressions/sec (Anemoi-2-1, BLS12-381)",
            "value": value, "unit": "compressions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "Anemoi-2-1 over BLS12-381 basefield, 2^%d batched Jive compressions per GPU, "
                                   "inputs resident in HBM" % args.batch_log2,
                       "field": FIELD, "state_width": WIDTH, "batch_per_gpu": n, "parallelism": "shard%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes)",
                         "traffic_source": prof_src,
                         "kernel": "k_jive<bls12_381,2,2>", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": BYTES_PER_ITEM * n},
            "alu": {"bound": "valu", "modmul_per_s": MODMUL_PER_ITEM * n / (kernel_ms * 1e-3),
                    "valu_issue_frac_profiled": valu_busy,
                    "mad_cycle_frac_profiled": mad_frac,
                    "note": "384-bit Montgomery mul/sqr per second (reference chain count 9576 per compression). "
                            "The path is VALU-issue bound: in the profiled kernel (SQ_INSTS_VALU, profiles/) "
                            "v_mad_u64_u32 at 16 lanes per clock alone fills mad_cycle_frac_profiled of all SIMD "
                            "cycles, and all VALU instructions priced at their measured issue cost fill "
                            "valu_issue_frac_profiled; see DESIGN.md"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
