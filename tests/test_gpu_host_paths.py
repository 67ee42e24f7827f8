"""The host-pointer entry points that round 2 left outside the chunked pipeline -- the ragged sponge, batched
path verification (binary and arity 4) and the run-time instances -- now go through rt::pipeline_staged
(csrc/runtime.h): chunks cut at message / item boundaries, three chunks resident on the device.  These
tests force MANY small chunks (options chunk_target_bytes, test_quantum: anemoi_set_option) so
that the three-slot ring wraps several times, in both staging modes, and compare every result with the
oracle (reference semantics: src/<f>/anemoi_*/hasher.rs:19-91 for the sponge, :86-92 for merge, :162-179 for
compress_k).  Bit-exact.  Run on the GPU box: pytest -m gpu.
"""
import os
import random

import numpy as np
import pytest

from conftest import FIELD_IDS, knobs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import anemoi_amd
    assert anemoi_amd.device_count() >= 1
    return anemoi_amd


@pytest.fixture(scope="module")
def R():
    import anemoi_ref
    return anemoi_ref


@pytest.mark.parametrize("staging", ["pinned", "direct"])
def test_ragged_sponge_through_many_chunks(A, oracle, staging):
    """~700 messages of 0..300 bytes plus a few far longer than the chunk target, cut into chunks of <= 4 KB of
    message bytes: every digest equals the oracle's hash of that message alone; the sharded form agrees."""
    rng = np.random.default_rng(77)
    for field, width in (("bn_254", 4), ("bls12_381", 2), ("jubjub", 2), ("bls12_377", 4)):
        fid, inst = FIELD_IDS.index(field), A.Anemoi(field, width)
        lens = [int(v) for v in rng.integers(0, 300, size=700)]
        for pos, ln in ((5, 9000), (64, 4097), (65, 0), (333, 12345), (699, 5000)):
            lens[pos] = ln
        msgs = [rng.integers(0, 256, size=n, dtype=np.uint8).tobytes() for n in lens]
        whole = inst.hash_ragged(msgs)                      # one chunk (default target)
        with knobs(chunk_target_bytes=4096, test_quantum=128, host_staging=staging):
            got = inst.hash_ragged(msgs)
            with knobs(virtual_devices=3):
                many = A.Anemoi(field, width, device=A.ALL_DEVICES).hash_ragged(msgs)
        assert (got == whole).all() and (many == whole).all(), (field, width)
        for i in list(range(0, 700, 7)) + [5, 64, 65, 333, 699]:
            assert (got[i] == oracle.hash_bytes(fid, width, msgs[i])).all(), (field, width, i, lens[i])
        # all-empty batch and a batch of one long message
        with knobs(chunk_target_bytes=4096, test_quantum=128, host_staging=staging):
            assert (inst.hash_ragged([b""] * 200) == 0).all()
            one = inst.hash_ragged([msgs[333]])
        assert (one[0] == whole[333]).all()


@pytest.mark.parametrize("staging", ["pinned", "direct"])
def test_path_verification_through_many_chunks(A, oracle, R, staging):
    """5 000 (leaf, index, path) triples of a depth-6 tree, every 13th one tampered with, verified in chunks of
    128 items: the accept / reject pattern is exactly the tampering pattern.  Binary and arity-4 trees."""
    rng = random.Random(5)
    for field, arity in (("jubjub", 2), ("bls12_381", 2), ("bn_254", 4), ("pallas", 4)):
        width = 2 if arity == 2 else 4
        fid, inst = FIELD_IDS.index(field), A.Anemoi(field, width)
        depth = 6 if arity == 2 else 3
        nleaf = arity ** depth
        I = R.Instance(field, width)
        leaves = oracle.ints_to_mont(fid, [rng.randrange(I.p) for _ in range(nleaf)])
        if arity == 2:
            levels = inst.merkle_tree(leaves, depth)
            path_of = lambda i: inst.merkle_path(levels, depth, i)
            verify = inst.merkle_verify_batch
        else:
            levels = inst.merkle_tree_arity4(leaves, depth)
            path_of = lambda i: inst.merkle_path_arity4(levels, depth, i)
            verify = inst.merkle_verify_arity4_batch
        root = levels[-1][0]
        all_paths = np.stack([path_of(i) for i in range(nleaf)])
        n = 5000
        idx = np.array([rng.randrange(nleaf) for _ in range(n)])
        sel, paths = leaves[idx].copy(), all_paths[idx].copy()
        bad = np.zeros(n, dtype=bool)
        for i in range(0, n, 13):
            bad[i] = True
            kind = (i // 13) % 3
            if kind == 0:
                sel[i, 0] ^= np.uint64(1)                                  # tampered leaf
            elif kind == 1:
                paths[i].reshape(-1)[rng.randrange(paths[i].size)] ^= np.uint64(4)   # tampered sibling
            else:
                idx[i] ^= 1                                                # right leaf, wrong slot
        whole = verify(sel, idx, paths, depth, root)
        with knobs(chunk_target_bytes=1, test_quantum=128, host_staging=staging):
            got = verify(sel, idx, paths, depth, root)
            with knobs(virtual_devices=3):
                sharded = A.Anemoi(field, width, device=A.ALL_DEVICES)
                many = (sharded.merkle_verify_batch if arity == 2 else sharded.merkle_verify_arity4_batch)(
                    sel, idx, paths, depth, root)
        assert (np.asarray(got, dtype=bool) == ~bad).all(), (field, arity)
        assert (np.asarray(whole, dtype=bool) == ~bad).all() and (np.asarray(many, dtype=bool) == ~bad).all()


@pytest.mark.parametrize("staging", ["pinned", "direct"])
def test_run_time_instances_through_many_chunks(A, R, oracle, staging):
    """A 3-column run-time instance over 1 500 states in chunks of 64 states: permutation (in place), Jive and
    the sponge equal the one-chunk run, and a sample equals the Python restatement of the reference's arms."""
    from test_gpu_generic import make_instance
    field, cols = "jubjub", 3
    gpu, ref, enc, dec = make_instance(A, R, oracle, field, cols, 3, seed=4242)
    rng = random.Random(9)
    w, n = 2 * cols, 1500
    st_i = [[rng.randrange(ref.p) for _ in range(w)] for _ in range(n)]
    st = np.stack([enc(s) for s in st_i])
    msgs = np.random.default_rng(3).integers(0, 256, size=(n, 100), dtype=np.uint8)
    whole_p, whole_j, whole_h = gpu.permutation_batch(st), gpu.compress_k_batch(st, 2), gpu.hash_batch(msgs, w - 1)
    with knobs(chunk_target_bytes=1, test_quantum=64 * cols, host_staging=staging):
        got_p, got_j, got_h = gpu.permutation_batch(st), gpu.compress_k_batch(st, 2), gpu.hash_batch(msgs, w - 1)
    assert (got_p == whole_p).all() and (got_j == whole_j).all() and (got_h == whole_h).all()
    for i in range(0, n, 97):
        assert dec(got_p[i]) == ref.permutation(list(st_i[i]))
        assert dec(got_j[i]) == ref.compress_k(st_i[i], 2)
    # overlapping input / output of the run-time Jive is refused like the fixed-instance one (partial overlap too)
    flat = np.zeros(n * w * ref.limbs + 64, dtype=np.uint64)
    from anemoi_amd import _lib
    inp, outp = flat[:n * w * ref.limbs], flat[8:8 + n * cols * ref.limbs]
    rc = A.lib.anemoi_generic_jive_compress_k_batch(gpu._inst_ref(), 2, inp.ctypes.data_as(_lib._u64p),
                                                    outp.ctypes.data_as(_lib._u64p), n, 0)
    assert rc == -3


def test_sponge_segments_without_pinned_staging(A, oracle):
    """host_staging=direct: the segment-fed sponge uploads each segment as one strided copy straight from
    the caller's memory instead of gathering into pinned staging (it used to ignore the knob, and to fail when
    pinning failed).  Same digests as the oracle."""
    rng = np.random.default_rng(5)
    fid, inst = FIELD_IDS.index("bn_254"), A.Anemoi("bn_254", 4)
    n, unit = 50, 3 * inst.chunk
    msgs = rng.integers(0, 256, size=(n, 7 * unit + 11), dtype=np.uint8)
    with knobs(sponge_segment_bytes=n * unit * 2, host_staging="direct"):
        got = inst.hash_batch(msgs)
    assert (got == oracle.hash_bytes_batch(fid, 4, msgs, threads=4)).all()


def test_unsorted_ragged_batch_is_bucketed_by_length_inside_the_library(A, oracle):
    """A long-tailed, UNSORTED ragged batch (most messages short, a few long) is staged by descending block count and
    the digests scattered back: every digest equals the oracle's hash of that message alone, in the caller's order --
    in one chunk, through many small chunks, and sharded."""
    rng = np.random.default_rng(4242)
    for field, width in (("bn_254", 4), ("jubjub", 2), ("bls12_381", 2)):
        fid, inst = FIELD_IDS.index(field), A.Anemoi(field, width)
        n = 1500
        lens = np.where(rng.integers(0, 12, size=n) == 0, rng.integers(900, 2500, size=n), rng.integers(0, 120, size=n))
        msgs = [rng.integers(0, 256, size=int(v), dtype=np.uint8).tobytes() for v in lens]
        got = inst.hash_ragged(msgs)
        with knobs(chunk_target_bytes=8192, test_quantum=128):
            many = inst.hash_ragged(msgs)
            with knobs(virtual_devices=3):
                shard = A.Anemoi(field, width, device=A.ALL_DEVICES).hash_ragged(msgs)
        assert (got == many).all() and (got == shard).all(), (field, width)
        for i in list(range(0, n, 11)) + [int(np.argmax(lens)), int(np.argmin(lens))]:
            assert (got[i] == oracle.hash_bytes(fid, width, msgs[i])).all(), (field, width, i, int(lens[i]))


def test_few_long_messages_stay_segmented_and_bound_the_device_footprint(A, oracle):
    """A batch that is SMALL by count (<= the cooperative sponge's cut-off) but LARGE by bytes -- 4 096 messages of
    64 KiB, 256 MiB -- must stream through the segment-fed sponge (three resident segments of ~24 MiB plus the carried
    states), not ride the single-launch latency route with the whole batch on the device and in pinned staging (round
    3's routing did, see capi.hip `latency_batch`).  Asserted on what the device actually holds after the call
    (hipMemGetInfo before / after, from a released library), and on sampled digests against the oracle."""
    import torch
    field, width = "bn_254", 4
    fid, inst = FIELD_IDS.index(field), A.Anemoi(field, width)
    n, ln = 4096, 64 * 1024 + 17          # a ragged tail in the last block of every message
    total = n * ln
    assert n <= 4 * 1024 and total > (200 << 20)
    msgs = np.random.default_rng(77).integers(0, 256, size=(n, ln), dtype=np.uint8)
    inst.hash_batch(msgs[:2, :100])       # (the first launch of a process loads the library's code objects onto the device:
    A.release(0)                          #  ~130 MB that are not this call's footprint -- the test must also pass on its own)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    got = inst.hash_batch(msgs)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(0)
    held = free0 - free1                   # the lane keeps its buffers until anemoi_release: this is the call's footprint
    assert 0 < held < (120 << 20), "device footprint %.1f MiB for a %.0f MiB batch" % (held / 2**20, total / 2**20)
    for i in (0, 1, n // 2, n - 1):
        assert (got[i] == oracle.hash_bytes(fid, width, msgs[i].tobytes())).all(), i
    A.release(0)


def test_mixed_entry_points_from_many_threads_while_the_cut_offs_change(A, oracle):
    """Seven host threads, each hammering a different host-pointer entry point (small and chunked Jive batches, both
    sponges, both ragged forms, a Merkle root, a permutation) on different fields, while an eighth flips the
    kernel-selection cut-offs back and forth: every call must return what the same call returned single-threaded
    (the options choose kernels, never results; lanes, pools and constant tables are shared state -- and are rebuilt by
    the threads' first calls, all at once, after an anemoi_release).  One result of each
    kind is also checked against the oracle."""
    import threading
    import time
    rng = np.random.default_rng(2026)
    el = lambda *shape: rng.integers(0, 1 << 60, size=shape, dtype=np.uint64)
    jub, bn, bls, ves = (A.Anemoi(f, w) for f, w in (("jubjub", 2), ("bn_254", 4), ("bls12_381", 2), ("vesta", 4)))
    msgs = [rng.integers(0, 256, size=int(k), dtype=np.uint8).tobytes() for k in rng.integers(0, 500, size=700)]
    emsgs = [el(int(k), 4) for k in rng.integers(0, 9, size=500)]
    jobs = {
        "jive small": (lambda s=el(300, 2, 4): jub.compress_batch(s)),
        "jive chunked": (lambda s=el(200000, 2, 6): bls.compress_batch(s)),
        "sponge bytes": (lambda m=rng.integers(0, 256, size=(900, 333), dtype=np.uint8): bn.hash_batch(m)),
        "sponge elements": (lambda e=el(3000, 7, 4): ves.hash_field_batch(e)),
        "ragged bytes": (lambda: jub.hash_ragged(msgs)),
        "ragged elements": (lambda: bn.hash_field_ragged(emsgs)),
        "merkle root": (lambda lv=el(1 << 12, 4): jub.merkle_root(lv, 12)),
        "permutation": (lambda s=el(5000, 4, 4): ves.permutation_batch(s)),
    }
    want = {k: np.array(f()) for k, f in jobs.items()}
    fid = FIELD_IDS.index("jubjub")
    assert (want["ragged bytes"][5] == oracle.hash_bytes(fid, 2, msgs[5])).all()
    assert (want["ragged elements"][7] == oracle.hash_field(FIELD_IDS.index("bn_254"), 4, emsgs[7])).all()
    errs, stop = [], threading.Event()

    def worker(name):
        try:
            for _ in range(4):
                if not (np.array(jobs[name]()) == want[name]).all():
                    errs.append(name)
        except Exception as e:      # noqa: BLE001 -- any error in a thread is a failure of the test
            errs.append("%s: %r" % (name, e))

    def flipper():
        keys = ("coop2d_max", "coop2d43_max", "coop4_max", "coop43_max", "coop_sponge_max")
        saved = {k: A.get_option(k) for k in keys}
        i = 0
        while not stop.is_set():
            for k in keys:
                A.set_option(k, 0 if i % 2 == 0 else saved[k])
            i += 1
            time.sleep(0.002)
        for k in keys:
            A.set_option(k, saved[k])

    A.release(0)        # everything the library holds is gone: the threads' FIRST calls rebuild lanes and constant tables concurrently
    fl = threading.Thread(target=flipper)
    fl.start()
    ths = [threading.Thread(target=worker, args=(k,)) for k in jobs]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    stop.set()
    fl.join()
    assert not errs, errs


def test_repeated_calls_do_not_grow_the_device_or_host_footprint(A):
    """The lanes keep their buffers between calls (the steady state allocates nothing): after one pass over a mix of
    entry points at their largest sizes, 60 more passes at equal or smaller sizes must not take another byte of device
    memory (hipMemGetInfo) nor grow the process (resident set) -- a per-call hipMalloc / pinned allocation that is
    never returned would show as a slope; anemoi_release then gives the device memory back."""
    import torch
    import psutil
    rng = np.random.default_rng(3)
    el = lambda *shape: rng.integers(0, 1 << 60, size=shape, dtype=np.uint64)
    jub, bn = A.Anemoi("jubjub", 2), A.Anemoi("bn_254", 4)
    st, msgs, lv = el(300000, 2, 4), rng.integers(0, 256, size=(20000, 500), dtype=np.uint8), el(1 << 14, 4)
    rag = [rng.integers(0, 256, size=int(k), dtype=np.uint8).tobytes() for k in rng.integers(0, 3000, size=3000)]
    emsgs = [el(int(k), 4) for k in rng.integers(0, 12, size=3000)]

    def one_pass(scale):
        jub.compress_batch(st[:len(st) // scale])
        bn.hash_batch(msgs[:len(msgs) // scale])
        jub.hash_ragged(rag[:len(rag) // scale])
        bn.hash_field_ragged(emsgs[:len(emsgs) // scale])
        jub.merkle_root(lv, 14)
        levels = jub.merkle_tree(lv[:1 << 10], 10)
        path = jub.merkle_path(levels, 10, 5)
        assert jub.merkle_verify_batch(lv[5:6], np.array([5], dtype=np.uint64), path[None], 10, levels[-1][0]).all()

    one_pass(1)
    one_pass(1)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    rss0 = psutil.Process().memory_info().rss
    for i in range(60):
        one_pass(1 + i % 3)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(0)
    rss1 = psutil.Process().memory_info().rss
    assert free0 - free1 <= 0, "device memory grew by %d bytes over 60 passes" % (free0 - free1)
    assert rss1 - rss0 < (64 << 20), "the process grew by %.1f MiB over 60 passes" % ((rss1 - rss0) / 2**20)
    assert A.lib.anemoi_release(0) == 0
    free2, _ = torch.cuda.mem_get_info(0)
    assert free2 > free1
    one_pass(3)                                   # ... and the library works on after a release
