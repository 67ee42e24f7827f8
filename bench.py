#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: Jive 2-to-1 compressions/s (Anemoi-2-1, BLS12-381).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (anemoi_jive_compress_k_dev, k = 2) over one batch of seeded
synthetic states that are already resident in HBM.
  N = 1  config 2 of BASELINE.json: 2^20 states (anemoi_amd/synth.py, seed 0xA9E30102).
  N > 1  config 4: the 2^24-state batch (seed 0xA9E30104) cut into 8 contiguous shards of 2^21; rank r
         processes shard r on its own GPU (N = 8 is config 4 whole; N = 2, 4 run its first N shards).  Items
         are independent, so there is no data-path collective; torch.distributed
         (RCCL) carries only the barrier and the max-over-ranks of the elapsed time.  Per-GPU work is
         fixed for every N > 1 ("weak" scaling); N = 1 runs half of that (config 2), at the same rate.
`value` = all ranks' items / max-over-ranks time.
  ANEMOI_BENCH_FORCE_DIST=1 makes a WORLD_SIZE = 1 launch (`torch.distributed.run --nproc-per-node 1`) take the N > 1
  code path: the RCCL process group with `device_id`, the communicator-creating barrier, the max-over-ranks on a device
  tensor, the all-ranks failure flag, shard 0 of config 4 -- so that the control plane of the first real scaling run has
  executed on the one-GPU boxes the builder gets (RCCL refuses two ranks on one device; it accepts one).

The line proves itself: after the timed loop every rank compares the outputs still sitting in its output
buffer with the committed oracle goldens of that exact batch (tests/golden/cfg_full.json, minted by
tools/mint_cfg_goldens.py from the CPU oracle): a strided sample item by item, and the SHA-256 of the
whole buffer.  A mismatch exits non-zero and prints no line.  No oracle code runs in or before the timed
region.

PyTorch is plumbing only (device buffers, the stream, torch.distributed for the barrier/max); the
work is done by libanemoi_mi355x.so through its C-ABI.

Also printed in the same JSON line:
  roofline      algorithmic bytes (144 B per compression: 96 in + 48 out) / the kernel's average
                launch duration measured with HIP events on the launch stream, against 8 TB/s HBM.
                The path is VALU-bound by ~4 orders of magnitude (9 576 384-bit modmul per 144 B), so
                this fraction is tiny by construction; `alu` carries the meaningful efficiency figure.
                `traffic` comes from the committed rocprofv3 --pmc passes of this command and is
                reported only when that profile was taken from the kernel sources being run
                (csrc hash match); otherwise it is null and `traffic_stale` is true.
  alu           v_mad_u64_u32 lane-operations per second / (1024 SIMDs x 16 lanes x clock) -- at the chip's 2.4 GHz maximum
                (`frac`) and at the clock MEASURED during the timed steps by a sampler that sits beside the kernel
                (`clock_GHz_measured*`, `frac_at_measured_clock`, `kernel_Mcycles_slowest_xcd`: what separates a slow BOX from a
                slow BUILD in the line itself) -- and the same quantity for a fixed probe kernel run right after the
                timed steps (anemoi_probe_issue_rate: a full grid, three wavefronts per SIMD, of dependent v_mad_u64_u32 chains):
                `probe_lane_mad_per_s`, the clock the chip held during it (`probe_clock_GHz`, s_memtime / s_memrealtime), and the
                same for a chain of the generated squaring, whose instruction mix is the kernel's (`probe_sqr_*`);
                `frac_of_sqr_probe` = the kernel's multiply-add rate / that probe's.  The boxes of the pool differ by several
                per cent under this load; frac_of_sqr_probe does not.
  cpu_baseline  the pinned C oracle ("port": same algorithm, u64-limb CIOS like arkworks) timed on a
                bounded sample on this box's host cores, all threads and one thread (rank 0, N = 1 only),
                with the CPU model; plus a probe for a Rust toolchain that could time the reference itself.
"""
import argparse
import hashlib
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "anemoi-rust_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

FIELD, WIDTH, LIMBS = "bls12_381", 2, 6
BYTES_PER_ITEM = 96 + 48          # 2 x 48 B in + 48 B out
MODMUL_PER_ITEM = 9576            # 21 rounds x (454 + 2), reference chain
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E peak
SIMDS, LANES_PER_CLK, MAX_GHZ = 1024, 16, 2.4   # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, max clock 2400 MHz


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def reference_toolchain_probe():
    """Could the reference itself (Rust + arkworks from crates.io, no Cargo.lock) be timed here?  Probe only."""
    cargo = shutil.which("cargo")
    rustc = shutil.which("rustc")
    vendored = any(os.path.isdir(os.path.expanduser(p)) for p in ("~/.cargo/registry", "/usr/local/cargo/registry"))
    return {"cargo": cargo, "rustc": rustc, "vendored_registry": vendored,
            "usable": bool(cargo and rustc and vendored)}


def cpu_baseline(synth, budget_s=10.0):
    """Oracle timed on a bounded sample of the same workload (rank 0, N = 1): all usable threads, then one."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    path = None
    try:  # prefer a -march=native build for this host
        path = orc.build(native=True, out=os.path.join("/tmp", "liboracle_native_%d.so" % os.getpid()))
    except Exception:
        path = None
    oracle = orc.Oracle(path)
    cores = usable_cores()

    def timed(threads, budget):
        st = synth.states(FIELD, WIDTH, 0xC0, 0, 64 * threads)
        t0 = time.perf_counter()
        oracle.compress_batch(0, WIDTH, st, threads=threads)
        rate = len(st) / (time.perf_counter() - t0)
        n = int(max(64 * threads, min(rate * budget, 1 << 20)))
        st = synth.states(FIELD, WIDTH, 0xC1, 0, n)
        t0 = time.perf_counter()
        oracle.compress_batch(0, WIDTH, st, threads=threads)
        dt = time.perf_counter() - t0
        return n / dt, n, dt

    rate, n, dt = timed(cores, budget_s)
    rate1, n1, dt1 = timed(1, budget_s / 3)
    return {"value": rate, "unit": "compressions/s", "cores": cores, "kind": "port",
            "single_thread": rate1, "cpu_model": cpu_model(), "compiler": orc.compiler_description(),
            "sample": "%d BLS12-381 Anemoi-2-1 Jive compressions on %d threads in %.1f s; %d on 1 thread in %.1f s; "
                      "C oracle (u64 CIOS Montgomery, the arkworks algorithm restated), not the reference binary"
                      % (n, cores, dt, n1, dt1),
            "reference_binary": reference_toolchain_probe()}


def rank_shard(rank, world, batch_log2=None, sharded=None):
    """(config name, config, first item, item count) of `rank` out of `world`: config 2 alone on one GPU,
    contiguous 2^21-item shards of config 4 otherwise (`sharded`: world > 1, or forced by ANEMOI_BENCH_FORCE_DIST)."""
    from anemoi_amd import synth
    sharded = (world > 1) if sharded is None else sharded
    cfg_name, cfg = ("cfg4", synth.CFG4) if sharded else ("cfg2", synth.CFG2)
    lg = batch_log2 if batch_log2 is not None else (21 if sharded else 20)
    n = 1 << lg
    first = rank * n
    if first + n > cfg["n"]:
        raise SystemExit("rank %d: items [%d, %d) are beyond %s's %d" % (rank, first, first + n, cfg_name, cfg["n"]))
    return cfg_name, cfg, first, n


def verify_against_golden(out_host, golden, first, n, check_sha=True):
    """out_host: (n, 6) uint64 outputs of items [first, first + n) of the golden's batch.  Returns (number of
    sampled items compared, whether the whole-buffer SHA-256 was checked and equal, error text or None)."""
    stride = golden["sample_stride"]
    checked = 0
    for j, hx in enumerate(golden["sample"]):
        idx = j * stride
        if first <= idx < first + n:
            want = np.frombuffer(bytes.fromhex(hx), dtype=np.uint64)
            if not (out_host[idx - first] == want).all():
                return checked, False, "output of item %d differs from the oracle golden" % idx
            checked += 1
    sha_ok = None
    digest = hashlib.sha256(np.ascontiguousarray(out_host).tobytes()).hexdigest() if check_sha else None
    if not check_sha:
        pass
    elif first == 0 and n == golden["n"]:
        sha_ok = digest == golden["sha256"]
    elif "shard_sha256" in golden and n * golden["shards"] == golden["n"] and first % n == 0:
        sha_ok = digest == golden["shard_sha256"][first // n]
    if sha_ok is False:
        return checked, False, "SHA-256 of the output buffer differs from the oracle golden"
    if checked == 0:
        return 0, False, "no golden sample falls inside this rank's items; nothing verified"
    return checked, bool(sha_ok), None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-log2", type=int, default=None,
                    help="items per GPU (default: 20 for N = 1 = config 2, 21 for N > 1 = config 4's shard)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock-sampler", action="store_true",
                    help="do not sample the shader clock beside the timed steps (the counter passes of tools/collect_profiles.sh: "
                         "rocprofv3 --pmc serialises kernels, and the sampler's few reads would count as the kernel's traffic)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    # one process per GPU; ANEMOI_BENCH_BACKEND=gloo lets the N > 1 path be rehearsed on a box with
    # fewer GPUs than ranks (ranks then share devices round-robin; timing is meaningless there).
    # There is no silent fallback: if RCCL fails the run fails.
    backend = os.environ.get("ANEMOI_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    local_rank = local_rank % max(ndev, 1) if backend == "gloo" else local_rank
    torch.cuda.set_device(local_rank)
    dist, audit = None, None
    # the N > 1 code path at WORLD_SIZE = 1 (see the module docstring): one rank is all RCCL accepts on a one-GPU box
    distributed = world > 1 or os.environ.get("ANEMOI_BENCH_FORCE_DIST") == "1"
    if distributed:
        import torch.distributed as dist
        from anemoi_amd.shard import ControlPlaneAudit
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        audit = ControlPlaneAudit(dist).install()          # every tensor that crosses torch.distributed is counted
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            dist.barrier()  # create the RCCL communicator now, outside the timed region
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import anemoi_amd as A  # after torch: binds to the same HIP runtime
    from anemoi_amd import buildinfo, synth
    from anemoi_amd.shard import max_over_ranks
    fid = A.field_id(FIELD)
    cfg_name, cfg, first, n = rank_shard(rank, world, args.batch_log2, distributed)   # contiguous shards of the config's batch
    lg = n.bit_length() - 1
    dev = torch.device("cuda", local_rank)
    host = synth.states(FIELD, WIDTH, cfg["seed"], first, n)
    d_in = torch.from_numpy(host.view(np.int64).reshape(-1)).to(dev)
    d_out = torch.zeros(n * LIMBS, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()
    rc = A.lib.anemoi_init(local_rank, fid, WIDTH)   # constant tables up front: the steps only launch
    if rc != 0:
        raise A.AnemoiError(rc, A.lib.anemoi_last_error().decode())

    def step():
        rc = A.lib.anemoi_jive_compress_k_dev(fid, WIDTH, 2, d_in.data_ptr(), d_out.data_ptr(), n, stream.cuda_stream)
        if rc != 0:
            raise A.AnemoiError(rc, A.lib.anemoi_last_error().decode())

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # Everything the timed region needs is made BEFORE the warm-up, and the warm-up writes into a buffer of its own: between
    # the warm-up's last step and the first timed one the GPU then idles only for the barrier and the sampler's start.  A gap
    # of 5 ms makes the first step 3 % slower (the chip re-enters full load at a lower clock), 1 ms 0.5 %, 0.2 ms nothing
    # (profiles/r06/first_step_clock.txt); the zero-fill + second synchronise + allocations that used to sit here were ~ms.
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    # the clock the chip holds DURING the timed steps: 16 single-wavefront sampler workgroups on a stream of their own, asleep
    # between samples (anemoi_clock_sampler_*); started and stamped before the clock starts, stopped by the device itself
    # behind the last step
    sampler = None if args.no_clock_sampler else A.ClockSampler(dev)
    d_warm = torch.empty_like(d_out)       # the check below must see what the TIMED steps wrote: d_out stays zero until then

    def warm_step():
        rc = A.lib.anemoi_jive_compress_k_dev(fid, WIDTH, 2, d_in.data_ptr(), d_warm.data_ptr(), n, stream.cuda_stream)
        if rc != 0:
            raise A.AnemoiError(rc, A.lib.anemoi_last_error().decode())

    for _ in range(max(args.warmup, 0)):
        warm_step()
    barrier()
    if sampler:
        sampler.start(stream)
    torch.cuda.current_stream().synchronize()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(stream)
        step()
        b.record(stream)
    if sampler:
        sampler.finish(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    clk_mean, clk_min, clk_max, clk_groups = sampler.read() if sampler else (0.0, 0.0, 0.0, 0)
    if dist is not None:
        dist.barrier()
    elapsed = max_over_ranks(elapsed, dist, dev if backend == "nccl" else None)
    step_ms = [a.elapsed_time(b) for a, b in evs]
    kernel_ms = sum(step_ms) / max(len(step_ms), 1)
    # what THIS GPU delivers of the kernels' instruction right now (outside the timed steps, the chip still under load)
    # (median of three runs of the probes: the squaring probe's rate wanders by +-1 % from run to run on one box)
    probes = sorted((A.probe_issue_rate(local_rank) for _ in range(3)), key=lambda p: p[2])
    probe_rate, probe_ghz, probe_sqr_rate, probe_sqr_ghz = probes[1]
    probe_min = -max_over_ranks(-probe_sqr_rate, dist, dev if backend == "nccl" else None)   # the slowest rank's GPU

    # ---- the timed output against the oracle goldens of this exact batch (every rank checks its shard)
    with open(os.path.join(ROOT, "tests", "golden", "cfg_full.json")) as f:
        golden = json.load(f)[cfg_name]
    assert golden["seed"] == cfg["seed"] and golden["n"] == cfg["n"]
    out_host = d_out.cpu().numpy().view(np.uint64).reshape(n, LIMBS)
    if os.environ.get("ANEMOI_BENCH_TEST_CORRUPT_RANK") == str(rank):
        # test hook (tests/test_gpu_configs.py): one wrong bit in this rank's output must make EVERY rank exit non-zero
        out_host = out_host.copy()
        out_host[n // 2, 1] ^= np.uint64(1)
    checked, sha_ok, err = verify_against_golden(out_host, golden, first, n)
    if err:
        sys.stderr.write("bench.py: rank %d: %s\n" % (rank, err))
    # every rank learns whether any rank failed, and all leave together (no line is printed)
    if max_over_ranks(1.0 if err else 0.0, dist, dev if backend == "nccl" else None) != 0.0:
        if dist is not None:
            dist.destroy_process_group()
        raise SystemExit(3)

    if rank == 0:
        # HBM traffic comes from the committed rocprofv3 --pmc passes of this same command
        # (profiles/rNN/pmc_k_jive.json): counters cannot be read from inside the process.  It is reported
        # only if that profile was taken from the kernel sources being run.
        csrc = buildinfo.csrc_sha256()
        traffic, prof_src, prof_clock, stale, valu_per_item = None, None, None, None, None
        try:
            prof_dirs = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles"))
                               if d.startswith("r") and os.path.exists(os.path.join(ROOT, "profiles", d, "pmc_k_jive.json")))
            prof_src = os.path.join("profiles", prof_dirs[-1], "pmc_k_jive.json")
            pmc = json.load(open(os.path.join(ROOT, prof_src)))
            stale = pmc.get("csrc_sha256") != csrc
            if not stale:
                prof_clock = pmc["derived"]["clock_GHz"]
                valu_per_item = pmc["derived"]["valu_wave_instructions_per_wavefront"]  # = per lane = per compression
                if lg == 20:
                    traffic = pmc["derived"]["hbm_traffic_bytes_per_launch"]
        except Exception:
            prof_src, stale = None, None
        total_items = n * world * args.steps
        value = total_items / elapsed
        achieved = BYTES_PER_ITEM * n / (kernel_ms * 1e-3) / 1e9
        mad_per_item = buildinfo.bls12_381_mad_per_compression()
        lay, (sq_r, mu_r) = buildinfo.bls12_381_limb_layout(), buildinfo.bls12_381_products_per_round()
        lane_mad_per_s = mad_per_item * n / (kernel_ms * 1e-3)
        peak_lane_ops = SIMDS * LANES_PER_CLK * MAX_GHZ * 1e9   # 16 lanes per SIMD per clock at the chip's MAXIMUM clock
        out = {
            "metric": "Jive 2-to-1 compressions/sec (Anemoi-2-1, BLS12-381)",
            "value": value, "unit": "compressions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": ("BASELINE config 2: Anemoi-2-1 over BLS12-381 basefield, 2^%d batched Jive compressions "
                                    "on one GPU, inputs resident in HBM" % lg) if not distributed else
                                   ("BASELINE config 4: Anemoi-2-1 over BLS12-381, 2^24-state batch in 8 contiguous shards, "
                                    "rank r runs shard r (2^%d items per GPU, %d of 8 shards), no collective on the data path"
                                    % (lg, world)),
                       "field": FIELD, "state_width": WIDTH, "batch_per_gpu": n, "parallelism": "shard%d" % world,
                       "seed": cfg["seed"], "control_plane": (backend if distributed else "none"),
                       # everything that crossed torch.distributed in this run, counted by a wrapper around the module's
                       # collectives (anemoi_amd/shard.py: ControlPlaneAudit): barriers and 8-byte scalars, nothing else
                       "control_plane_traffic": (audit.report() if audit else None)},
            "verified": {"against": "tests/golden/cfg_full.json:%s (CPU oracle)" % cfg_name, "items_compared": checked,
                         "sha256_of_all_outputs": sha_ok, "ranks": world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_stale": stale,
                         "traffic_unit": "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes)",
                         "traffic_source": prof_src, "csrc_sha256": csrc,
                         "kernel": "k_jive<bls12_381,2,2>", "kernel_ms": kernel_ms,
                         "kernel_ms_each_step": [round(v, 3) for v in step_ms[:32]],   # (rank 0's; the first step behind the barrier: see above)
                         "algorithmic_bytes_per_launch": BYTES_PER_ITEM * n},
            "alu": {"bound": "valu", "modmul_per_s": MODMUL_PER_ITEM * n / (kernel_ms * 1e-3),
                    "mad_per_item": mad_per_item, "lane_mad_per_s": lane_mad_per_s,
                    # THE figure: multiply-add lane-operations against 1024 SIMDs x 16 lanes x the 2.4 GHz MAXIMUM clock
                    # (MI355X_MICROARCH.md).  The clock a run actually holds cannot be read from inside the process and
                    # differs from box to box (2.27-2.40 GHz under this load), so a fraction quoted at a borrowed clock
                    # can overstate; at the maximum clock it cannot, on any box of the pool.
                    "peak_lane_mad_per_s": peak_lane_ops, "clock_GHz": MAX_GHZ, "clock_source": "maximum engine clock",
                    "frac": lane_mad_per_s / peak_lane_ops,
                    # MEASURED in this run: the shader clock the chip held during the timed steps (mean / slowest / fastest of
                    # 16 sampler workgroups spread over the XCDs; s_memtime against the 100 MHz s_memrealtime).  The boxes of
                    # the pool differ in exactly this, by up to 3 %.  `frac_at_measured_clock` = the same multiply-add rate
                    # against 1024 SIMDs x 16 lanes x the MEAN measured clock; `kernel_Mcycles_slowest_xcd` = kernel time x the
                    # SLOWEST XCD's clock -- workgroups are dealt round-robin to the XCDs, the slowest one finishes last --
                    # which is the box-independent cost of the build (226.7-231.2 on boxes whose `value` ranged 9.30-10.79
                    # M/s: profiles/r05/): a slower box moves `clock_GHz_measured*`, a slower build moves
                    # `kernel_Mcycles_slowest_xcd` and `frac_at_measured_clock`.
                    "clock_GHz_measured": clk_mean, "clock_GHz_measured_slowest_xcd": clk_min, "clock_GHz_measured_fastest_xcd": clk_max,
                    "clock_sampler_groups": clk_groups,
                    "frac_at_measured_clock": (lane_mad_per_s / (SIMDS * LANES_PER_CLK * clk_mean * 1e9)) if clk_mean else None,
                    "kernel_Mcycles_slowest_xcd": (kernel_ms * clk_min) if clk_min else None,
                    # box or build?  Two fixed probe kernels on THIS GPU, right after the timed steps: bare dependent
                    # multiply-add chains (`probe_*`: what the instruction can do here, and the clock the chip holds for it)
                    # and a chain of the generated BLS12-381 squaring (`probe_sqr_*`: the instruction MIX and power draw of the
                    # benchmarked kernel, which is 80 % squarings).  A slower box moves `value` and `probe_sqr_lane_mad_per_s`
                    # together; a slower build moves `frac_of_sqr_probe` -- the figure to compare across boxes and rounds.
                    "probe_lane_mad_per_s": probe_rate, "probe_clock_GHz": probe_ghz,
                    "probe_frac_of_peak": probe_rate / peak_lane_ops, "frac_of_probe": lane_mad_per_s / probe_rate,
                    "probe_sqr_lane_mad_per_s": probe_sqr_rate, "probe_sqr_clock_GHz": probe_sqr_ghz,
                    "frac_of_sqr_probe": lane_mad_per_s / probe_sqr_rate,
                    "probe_sqr_lane_mad_per_s_min_over_ranks": probe_min,
                    # secondary: the same ratio at the clock of the committed profile run (GRBM_GUI_ACTIVE / duration
                    # on THAT box); indicative only
                    "frac_at_profile_clock": (lane_mad_per_s / (SIMDS * LANES_PER_CLK * prof_clock * 1e9)) if prof_clock else None,
                    "profile_clock_GHz": prof_clock,
                    # all VALU lane-instructions (SQ_INSTS_VALU per compression of the committed profile x this run's
                    # rate) against the same maximum-clock peak: must be <= 1 on every box; NOT clamped -- a value above
                    # 1 would mean the instruction count or the peak model is wrong and is flagged
                    "valu_instr_per_item": valu_per_item,
                    "valu_issue_frac": (valu_per_item * n / (kernel_ms * 1e-3) / peak_lane_ops) if valu_per_item else None,
                    # (> 1.005: half a per cent for the counter's own granularity; round 4's boxes land at 0.98-1.00, i.e.
                    # the vector ALUs issue in every cycle the maximum clock has)
                    "valu_issue_inconsistent": bool(valu_per_item and valu_per_item * n / (kernel_ms * 1e-3) / peak_lane_ops > 1.005),
                    "note": "v_mad_u64_u32 lane-operations per second (count per compression from the generated assembly and "
                            "exponent schedule: 21 rounds x (%d squarings x %d + %d multiplications x %d) + 5 x %d) against "
                            "1024 SIMDs x 16 lanes per clock at 2.4 GHz; the path is VALU-bound, see DESIGN.md"
                            % (sq_r, lay["sqr_mad"], mu_r, lay["mul_mad"], lay["mul_mad"])},
        }
        if not distributed and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(synth)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
