"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/anemoi_mi355x.h
declares, answers introspection calls, and rejects bad arguments with the documented codes BEFORE
touching a device (the reference's assert!s).  No compute calls here (no GPU in this container)."""
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
    assert A.lib.anemoi_abi_version() == 4
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
    with pytest.raises(A.AnemoiError):
        A.Anemoi("bls12_381", 3)
    with pytest.raises(A.AnemoiError):
        A.Anemoi("bls12_381", 2).compress(np.zeros((3, 6), dtype=np.uint64))  # wrong length


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
