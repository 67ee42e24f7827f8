"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/anemoi_mi355x.h
declares, answers introspection calls, and rejects bad arguments with the documented codes BEFORE
touching a device (the reference's assert!s).  No compute calls here (no GPU in this container)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import FIELD_IDS, ROOT


@pytest.fixture(scope="module")
def A():
    import anemoi_amd
    return anemoi_amd


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "anemoi_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(anemoi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(A):
    syms = declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(A.lib, s)]
    assert not missing, missing


def test_binding_covers_every_declared_symbol(A):
    from anemoi_amd import _lib
    assert sorted(_lib._SIGS) == declared_symbols()


def test_introspection(A, params):
    assert A.lib.anemoi_abi_version() == 500        # 100 x major + minor: the rule is in the header
    for fid, name in enumerate(FIELD_IDS):
        assert A.lib.anemoi_field_name(fid).decode() == name
        assert A.field_id(name) == fid
        fp = params[name]
        assert A.lib.anemoi_field_limbs(fid) == fp["u64_limbs"]
        assert A.lib.anemoi_field_chunk_bytes(fid) == fp["byte_chunk"]
        assert A.lib.anemoi_num_rounds(fid, 2) == fp["instances"]["anemoi_2_1"]["num_rounds"]
        assert A.lib.anemoi_num_rounds(fid, 4) == fp["instances"]["anemoi_4_3"]["num_rounds"]
    assert A.lib.anemoi_field_id(b"goldilocks") == -1
    assert A.lib.anemoi_field_limbs(7) == -1
    assert A.lib.anemoi_num_rounds(0, 3) == -2
    assert b"k" in A.lib.anemoi_strerror(-3)


def test_argument_errors_need_no_device(A):
    from anemoi_amd import _lib
    buf = np.zeros(64, dtype=np.uint64)
    p = buf.ctypes.data_as(_lib._u64p)
    L = A.lib
    assert L.anemoi_permutation_batch(9, 2, p, 1, 0) == -1            # unknown field
    assert L.anemoi_permutation_batch(0, 3, p, 1, 0) == -2            # bad width
    assert L.anemoi_permutation_batch(0, 2, None, 1, 0) == -3         # null pointer
    # the reference's compress_k asserts: 2-1 accepts only k == 2 (anemoi_2_1/hasher.rs:107),
    # 4-3 needs STATE_WIDTH % k == 0 and k % 2 == 0 (anemoi_4_3/hasher.rs:163-165)
    out = np.zeros(64, dtype=np.uint64)
    q = out.ctypes.data_as(_lib._u64p)
    for width, k in [(2, 4), (2, 1), (2, 3), (4, 3), (4, 1), (4, 8), (4, 0)]:
        assert L.anemoi_jive_compress_k_batch(0, width, k, p, q, 1, 0) == -3
    assert L.anemoi_merkle_root(4, p, 31, q, 0) == -3                 # depth out of range
    assert L.anemoi_hash_bytes_batch(0, 5, None, 0, 1, q, 0) == -2
    # round 5's entry points: bad arguments are refused before anything touches a device
    assert L.anemoi_clock_sampler_start_dev(None, 1 << 21, 2000, 1000, None) == -3
    assert L.anemoi_clock_sampler_start_dev(buf.ctypes.data, 16, 2000, 1000, None) == -3          # buffer too small
    big = np.zeros(L.anemoi_clock_sampler_bytes() // 8 + 1, dtype=np.uint64)
    for period_us, max_ms in [(9, 1000), (1000001, 1000), (2000, 0), (2000, 600001)]:
        assert L.anemoi_clock_sampler_start_dev(big.ctypes.data, big.nbytes, period_us, max_ms, None) == -3
    assert L.anemoi_clock_sampler_start_dev(big.ctypes.data + 4, big.nbytes - 4, 2000, 1000, None) == -3   # misaligned
    assert L.anemoi_clock_sampler_stop_dev(None, None) == -3
    assert L.anemoi_clock_sampler_wait_dev(None, 50, None) == -3
    assert L.anemoi_clock_sampler_wait_dev(big.ctypes.data, 0, None) == -3 and L.anemoi_clock_sampler_wait_dev(big.ctypes.data, 10001, None) == -3
    assert L.anemoi_clock_stamp_dev(None, None) == -3
    assert L.anemoi_clock_stamp_dev(buf.ctypes.data + 4, None) == -3
    assert L.anemoi_ragged_scratch_bytes(1000) == (4 + 65536 + 1000) * 4
    assert L.anemoi_hash_bytes_ragged_bucketed_dev(0, 2, p, 64, p, 4, q, None, 0, None) == -3         # no scratch
    assert L.anemoi_hash_bytes_ragged_bucketed_dev(0, 2, p, 64, p, 4, q, big.ctypes.data, 64, None) == -3   # scratch too small
    assert L.anemoi_hash_bytes_ragged_bucketed_dev(0, 2, p, 64, p, 4, q, big.ctypes.data + 2, big.nbytes - 2, None) == -3
    assert L.anemoi_hash_bytes_ragged_bucketed_dev(0, 2, p, 64, p, 0, q, big.ctypes.data, big.nbytes, None) == 0   # nothing to do
    assert L.anemoi_hash_bytes_ragged_bucketed_dev(0, 3, p, 64, p, 4, q, big.ctypes.data, big.nbytes, None) == -2
    # ... the same checks in front of the hash_field forms; decreasing element offsets
    assert L.anemoi_hash_field_ragged_bucketed_dev(0, 2, p, 64, p, 4, q, None, 0, None) == -3
    assert L.anemoi_hash_field_ragged_bucketed_dev(0, 2, p, 64, p, 4, q, big.ctypes.data, 64, None) == -3
    assert L.anemoi_hash_field_ragged_bucketed_dev(0, 2, p, 64, p, 0, q, big.ctypes.data, big.nbytes, None) == 0
    assert L.anemoi_hash_field_ragged_dev(7, 2, p, p, 4, q, None) == -1
    assert L.anemoi_hash_field_ragged_dev(0, 6, p, p, 4, q, None) == -2
    assert L.anemoi_hash_field_ragged_dev(0, 2, None, p, 4, q, None) == -3
    offs = np.array([0, 3, 2], dtype=np.uint64)
    assert L.anemoi_hash_field_ragged_batch(0, 2, p, offs.ctypes.data_as(_lib._u64p), 2, q, 0) == -3
    assert L.anemoi_hash_field_ragged_batch(0, 2, p, offs.ctypes.data_as(_lib._u64p), 0, q, 0) == 0
    assert L.anemoi_hash_field_ragged_batch(0, 2, None, np.array([0, 3], dtype=np.uint64).ctypes.data_as(_lib._u64p), 1, q, 0) == -3
    d = ctypes.c_double()
    assert L.anemoi_probe_issue_rate(0, None, ctypes.byref(d), ctypes.byref(d), ctypes.byref(d)) == -3
    with pytest.raises(A.AnemoiError):
        A.Anemoi("bls12_381", 3)
    with pytest.raises(A.AnemoiError):
        A.Anemoi("bls12_381", 2).compress(np.zeros((3, 6), dtype=np.uint64))  # wrong length


def test_clock_sampler_read_is_host_arithmetic(A):
    """anemoi_clock_sampler_read works on a HOST copy of the sampler's log: per workgroup the clock of every interval
    between two samples inside [t0, t1] (cycles per 10 ns tick / 10 = GHz), the mean of the middle 80 % of them; mean /
    min / max over the workgroups that have at least five such intervals.  A synthetic log: workgroup 0 at 2.0 GHz with
    one interval in which the cycle counter jumped, workgroup 1 at 2.4 GHz, workgroup 2 with too few samples, workgroup 3
    entirely outside the window, workgroup 4 with a corrupt count."""
    L = A.lib
    nbytes = L.anemoi_clock_sampler_bytes()
    groups, records = 16, 4096
    assert nbytes == 16 + groups * (8 + records * 16)
    raw = np.zeros(nbytes, dtype=np.uint8)

    def group(i):
        base = 16 + i * (8 + records * 16)
        return raw[base:base + 8].view(np.uint32), raw[base + 8:base + 8 + records * 16].view(np.uint64).reshape(records, 2)

    def fill(i, n, t_start, ticks, cycles_per_tick, jump_at=None):
        head, rec = group(i)
        head[0] = n
        t = t_start + ticks * np.arange(n, dtype=np.uint64)
        c = (np.arange(n, dtype=np.uint64) * np.uint64(ticks * cycles_per_tick)) + np.uint64(12345)
        if jump_at is not None:
            c[jump_at:] += np.uint64(10 ** 9)
        rec[:n, 0], rec[:n, 1] = t, c
    t0, t1 = 1_000_000, 1_000_000 + 200_000 * 60        # 60 samples of 2 ms inside the window
    fill(0, 64, t0 - 400_000, 200_000, 20, jump_at=30)  # 2.0 GHz, one spoilt interval
    fill(1, 64, t0 - 400_000, 200_000, 24)              # 2.4 GHz
    fill(2, 4, t0 + 200_000, 200_000, 22)               # three intervals: not used
    fill(3, 64, t1 + 200_000, 200_000, 30)              # after the window: not used
    fill(4, 64, t0, 200_000, 30)
    group(4)[0][0] = records + 1                        # a count no sampler can have written: not used
    v = [ctypes.c_double() for _ in range(3)]
    used = ctypes.c_int()
    args = [ctypes.byref(x) for x in v] + [ctypes.byref(used)]
    assert L.anemoi_clock_sampler_read(raw.ctypes.data, nbytes, t0, t1, *args) == 0
    assert used.value == 2
    assert abs(v[1].value - 2.0) < 1e-9 and abs(v[2].value - 2.4) < 1e-9 and abs(v[0].value - 2.2) < 1e-9
    assert L.anemoi_clock_sampler_read(raw.ctypes.data, nbytes, t1, t0, *args) == -3          # empty window
    assert L.anemoi_clock_sampler_read(raw.ctypes.data, nbytes - 1, t0, t1, *args) == -3
    assert L.anemoi_clock_sampler_read(raw.ctypes.data, nbytes, t1 + 10 ** 9, t1 + 2 * 10 ** 9, *args) == 0 and used.value == 0
    assert v[0].value == v[1].value == v[2].value == 0.0


def test_no_cpu_fallback_in_product():
    """The product tree must not reference the oracle (a CPU fallback would void parity claims)."""
    pkg = os.path.join(ROOT, "anemoi-rust_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "anemoi_oracle" not in text and "import orc" not in text, f
