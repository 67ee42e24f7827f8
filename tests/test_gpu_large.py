"""Batches whose buffers are larger than 4 GiB: every byte offset beyond 2^32 must be computed in 64 bits -- in the kernels
(item x bytes per item, offsets of ragged messages), in the chunk plans of the host pipelines and in the staging copies.
No other test crosses that line (the BASELINE configs stop at 1.5 GiB).

Method: a BASE batch of P items (P chosen so that neither P x item bytes nor 2^32 is a multiple of the other: an offset
that wrapped at 2^32 would land in the middle of a different item) is hashed on its own and checked against the oracle on a
sample; the big batch is the base repeated cyclically past 4 GiB, and EVERY one of its outputs must equal the base's
output at i mod P.  Host-pointer entry points (the chunked pipelines) and device-resident ones."""
import numpy as np
import pytest

from conftest import FIELD_IDS

pytestmark = pytest.mark.gpu
GIB4 = 1 << 32


@pytest.fixture(scope="module")
def A():
    import anemoi_amd
    return anemoi_amd


def cyc(base, n):
    """base repeated cyclically to n rows"""
    return np.resize(base, (n,) + base.shape[1:])


def test_jive_batch_of_more_than_4_gib(A, oracle):
    fid, inst = FIELD_IDS.index("jubjub"), A.Anemoi("jubjub", 2)
    rng = np.random.default_rng(41)
    P = (1 << 20) + 1
    base = rng.integers(0, 1 << 60, size=(P, 2, 4), dtype=np.uint64)          # limbs < 2^60: canonical elements
    want = inst.compress_batch(base)[:, 0]
    idx = rng.integers(0, P, size=48)
    assert (want[idx] == oracle.compress_batch(fid, 2, base[idx], threads=8).reshape(-1, 4)).all()
    n = GIB4 // 64 + (1 << 22) + 3                                            # 64 B per state: 4.25 GiB in, 2.1 GiB out
    big = cyc(base, n)
    assert big.nbytes > GIB4 and (GIB4 % (P * 64)) and ((P * 64) % 64 == 0)
    got = inst.compress_batch(big)[:, 0]
    del big
    assert got.shape == (n, 4)
    for lo in range(0, n, P):                                                  # period by period: no second 2 GiB array
        hi = min(lo + P, n)
        assert (got[lo:hi] == want[:hi - lo]).all(), "states %d ... %d" % (lo, hi)


def test_sponge_batches_of_more_than_4_gib_host_and_device(A, oracle):
    import torch
    fid, inst = FIELD_IDS.index("bn_254"), A.Anemoi("bn_254", 4)
    rng = np.random.default_rng(42)
    P, ln = 8193, 10240                                                        # 2^32 mod 10 240 = 4 096: a wrapped offset lands mid-message
    base = rng.integers(0, 256, size=(P, ln), dtype=np.uint8)
    want = inst.hash_batch(base)
    for i in (0, 1, P // 2, P - 1):
        assert (want[i] == oracle.hash_bytes(fid, 4, base[i].tobytes())).all()
    n = GIB4 // ln + 20011
    big = cyc(base, n)
    assert big.nbytes > GIB4
    got = inst.hash_batch(big)                                                 # host pointers: chunked copy / compute pipeline
    dev = torch.device("cuda", 0)
    d_msgs = torch.from_numpy(big).to(dev)
    del big
    d_out = torch.zeros(n * 4, dtype=torch.int64, device=dev)
    assert A.lib.anemoi_hash_bytes_dev(fid, 4, d_msgs.data_ptr(), ln, n, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    got_dev = d_out.cpu().numpy().view(np.uint64).reshape(n, 4)
    del d_msgs
    torch.cuda.empty_cache()
    for lo in range(0, n, P):
        hi = min(lo + P, n)
        assert (got[lo:hi] == want[:hi - lo]).all(), "host path, messages %d ... %d" % (lo, hi)
        assert (got_dev[lo:hi] == want[:hi - lo]).all(), "device path, messages %d ... %d" % (lo, hi)


def test_ragged_batch_with_offsets_beyond_4_gib_host_and_device(A, oracle):
    import torch
    fid, inst = FIELD_IDS.index("bn_254"), A.Anemoi("bn_254", 4)
    rng = np.random.default_rng(43)
    P = 4099
    lens = rng.integers(9000, 11000, size=P).astype(np.uint64)
    lens[7] = 0
    blob = rng.integers(0, 256, size=int(lens.sum()), dtype=np.uint8)
    boffs = np.zeros(P + 1, dtype=np.uint64)
    boffs[1:] = np.cumsum(lens)
    base_msgs = [blob[int(boffs[i]):int(boffs[i + 1])].tobytes() for i in range(P)]
    want = inst.hash_ragged(base_msgs)
    for i in (0, 7, P // 3, P - 1):
        assert (want[i] == oracle.hash_bytes(fid, 4, base_msgs[i])).all()
    reps = GIB4 // int(boffs[-1]) + 2
    n = reps * P
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(np.tile(lens, reps))
    big = np.tile(blob, reps)
    assert int(offs[-1]) == big.nbytes > GIB4
    from anemoi_amd import _lib
    got = np.empty((n, 4), dtype=np.uint64)
    assert A.lib.anemoi_hash_bytes_ragged_batch(fid, 4, big.ctypes.data_as(_lib._u8p), offs.ctypes.data_as(_lib._u64p), n,
                                                got.ctypes.data_as(_lib._u64p), 0) == 0
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    d_blob, d_offs = torch.from_numpy(big).to(dev), torch.from_numpy(offs.view(np.int64)).to(dev)
    del big
    need = A.lib.anemoi_ragged_scratch_bytes(n)
    d_scr = torch.empty(need, dtype=torch.uint8, device=dev)
    outs = []
    for bucketed in (0, 1):
        d_out = torch.zeros(n * 4, dtype=torch.int64, device=dev)
        if bucketed:
            rc = A.lib.anemoi_hash_bytes_ragged_bucketed_dev(fid, 4, d_blob.data_ptr(), d_blob.numel(), d_offs.data_ptr(), n, d_out.data_ptr(),
                                                             d_scr.data_ptr(), need, s)
        else:
            rc = A.lib.anemoi_hash_bytes_ragged_dev(fid, 4, d_blob.data_ptr(), d_offs.data_ptr(), n, d_out.data_ptr(), s)
        assert rc == 0
        torch.cuda.synchronize()
        outs.append(d_out.cpu().numpy().view(np.uint64).reshape(n, 4))
    del d_blob
    torch.cuda.empty_cache()
    for r in range(reps):
        for name, o in (("host", got), ("device, in order", outs[0]), ("device, bucketed", outs[1])):
            assert (o[r * P:(r + 1) * P] == want).all(), "%s path, repeat %d" % (name, r)


def test_merkle_root_over_more_than_4_gib_of_leaves(A, oracle):
    """2^28 Jubjub leaves = 8 GiB + the levels above: the root from ONE call (host leaves, level by level on the device)
    must equal the root over the 256 depth-20 subtree roots that 256 separate calls give (each on a 32 MiB slice, far
    from any 32-bit limit), the top eight levels finished by the oracle; the device entry point likewise (leaves and
    scratch resident: two 8 GiB buffers).  Leaves: 2^24 random elements repeated 16 times, the repeat number added to
    limb 0 -- every depth-20 subtree is different."""
    import torch
    fid, inst = FIELD_IDS.index("jubjub"), A.Anemoi("jubjub", 2)
    rng = np.random.default_rng(44)
    depth, sub, rep = 28, 20, 24
    base = rng.integers(0, 1 << 60, size=(1 << rep, 4), dtype=np.uint64)
    leaves = np.tile(base, (1 << (depth - rep), 1))
    leaves[:, 0] += np.repeat(np.arange(1 << (depth - rep), dtype=np.uint64), 1 << rep)
    assert leaves.nbytes == 2 * GIB4
    root = inst.merkle_root(leaves, depth)
    roots = np.stack([inst.merkle_root(leaves[i << sub:(i + 1) << sub], sub) for i in range(1 << (depth - sub))])
    assert len({r.tobytes() for r in roots}) == len(roots)
    assert (root == oracle.merkle_root(fid, roots, depth - sub)).all()
    dev = torch.device("cuda", 0)
    d_leaves = torch.from_numpy(leaves.view(np.int64).reshape(-1)).to(dev)
    del leaves
    d_scratch = torch.empty((1 << depth) * 4, dtype=torch.int64, device=dev)
    d_root = torch.zeros(4, dtype=torch.int64, device=dev)
    assert A.lib.anemoi_merkle_root_dev(fid, d_leaves.data_ptr(), depth, d_scratch.data_ptr(), d_root.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert (d_root.cpu().numpy().view(np.uint64) == root).all()
    del d_leaves, d_scratch
    torch.cuda.empty_cache()
