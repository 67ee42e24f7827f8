"""ctypes loader for the C oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this; the
product never does.  Builds the library with `make -C oracle` when it is missing.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
FIELD_IDS = ["bls12_381", "bls12_377", "bn_254", "ed_on_bls12_377", "jubjub", "pallas", "vesta"]

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u8p = ctypes.POINTER(ctypes.c_uint8)


_CLANG = "/opt/rocm/lib/llvm/bin/clang"


def _compiler():
    """The image's clang when present, else gcc without SLP vectorisation (gcc 11 -O3 turns the 6-limb
    CIOS loops into a 5x slower vector form; see oracle/Makefile)."""
    if os.path.exists(_CLANG):
        return [_CLANG, "-O3"]
    return ["gcc", "-O3", "-fno-tree-slp-vectorize"]


def compiler_description():
    cc = _compiler()
    try:
        ver = subprocess.run([cc[0], "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    except Exception:
        ver = "?"
    return "%s (%s)" % (" ".join(os.path.basename(c) if i == 0 else c for i, c in enumerate(cc)), ver)


def build(native=False, out=None):
    """Compile the oracle. native=True -> -march=native into `out` (for the CPU baseline on the GPU box)."""
    if not native:
        subprocess.check_call(["make", "-s", "-C", _HERE])
        return os.path.join(_HERE, "liboracle.so")
    out = out or os.path.join(_HERE, "liboracle_native.so")
    subprocess.check_call(_compiler() + ["-march=native", "-fPIC", "-std=gnu11", "-shared", "-o", out,
                                         os.path.join(_HERE, "anemoi_oracle.c"), "-lpthread"])
    return out


def load(path=None):
    path = path or os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    lib = ctypes.CDLL(path)
    sz = ctypes.c_size_t
    sig = {
        "orc_field_limbs": [ctypes.c_int],
        "orc_to_mont": [ctypes.c_int, _u64p, _u64p, sz],
        "orc_from_mont": [ctypes.c_int, _u64p, _u64p, sz],
        "orc_permutation": [ctypes.c_int, ctypes.c_int, _u64p],
        "orc_sbox_layer_state": [ctypes.c_int, ctypes.c_int, _u64p],
        "orc_compress_k": [ctypes.c_int, ctypes.c_int, _u64p, _u64p, ctypes.c_int],
        "orc_hash_field": [ctypes.c_int, ctypes.c_int, _u64p, sz, _u64p],
        "orc_hash_bytes": [ctypes.c_int, ctypes.c_int, _u8p, sz, _u64p],
        "orc_merge": [ctypes.c_int, ctypes.c_int, _u64p, _u64p, _u64p],
        "orc_digest_bytes": [ctypes.c_int, _u64p, _u8p],
        "orc_merkle_root": [ctypes.c_int, _u64p, ctypes.c_uint, _u64p],
        "orc_compress_batch": [ctypes.c_int, ctypes.c_int, ctypes.c_int, _u64p, _u64p, sz, ctypes.c_int],
        "orc_hash_bytes_batch": [ctypes.c_int, ctypes.c_int, _u8p, sz, sz, _u64p, ctypes.c_int],
        "orc_hash_field_batch": [ctypes.c_int, ctypes.c_int, _u64p, sz, sz, _u64p, ctypes.c_int],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, ctypes.c_int
    return lib


def _p64(a):
    return a.ctypes.data_as(_u64p)


def _p8(a):
    return a.ctypes.data_as(_u8p)


class Oracle:
    """numpy-facing wrapper. Elements are rows of `limbs` uint64 (Montgomery form, the C-ABI encoding)."""

    def __init__(self, path=None):
        self.lib = load(path)

    def limbs(self, field):
        return self.lib.orc_field_limbs(field)

    # ---- int <-> limb helpers (canonical integers on the Python side)
    def ints_to_mont(self, field, ints):
        L = self.limbs(field)
        flat = np.array([[(int(v) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(L)] for v in ints],
                        dtype=np.uint64).reshape(-1, L)
        out = np.empty_like(flat)
        assert self.lib.orc_to_mont(field, _p64(flat), _p64(out), len(flat)) == 0
        return out

    def mont_to_ints(self, field, arr):
        L = self.limbs(field)
        arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, L)
        out = np.empty_like(arr)
        assert self.lib.orc_from_mont(field, _p64(arr), _p64(out), len(arr)) == 0
        return [sum(int(out[r, i]) << (64 * i) for i in range(L)) for r in range(len(out))]

    # ---- single-state functions
    def permutation(self, field, width, state):
        st = np.ascontiguousarray(state, dtype=np.uint64).copy()
        assert self.lib.orc_permutation(field, width, _p64(st)) == 0
        return st

    def sbox_layer(self, field, width, state):
        st = np.ascontiguousarray(state, dtype=np.uint64).copy()
        assert self.lib.orc_sbox_layer_state(field, width, _p64(st)) == 0
        return st

    def compress_k(self, field, width, elems, k=2):
        L = self.limbs(field)
        e = np.ascontiguousarray(elems, dtype=np.uint64)
        out = np.empty((width // k, L), dtype=np.uint64)
        rc = self.lib.orc_compress_k(field, width, _p64(e), _p64(out), k)
        if rc:
            raise ValueError("orc_compress_k rc=%d" % rc)
        return out

    def hash_field(self, field, width, elems):
        L = self.limbs(field)
        e = np.ascontiguousarray(elems, dtype=np.uint64).reshape(-1, L)
        out = np.empty(L, dtype=np.uint64)
        assert self.lib.orc_hash_field(field, width, _p64(e), len(e), _p64(out)) == 0
        return out

    def hash_bytes(self, field, width, data):
        L = self.limbs(field)
        b = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
        out = np.empty(L, dtype=np.uint64)
        assert self.lib.orc_hash_bytes(field, width, _p8(b), len(data), _p64(out)) == 0
        return out

    def merge(self, field, width, left, right):
        L = self.limbs(field)
        a = np.ascontiguousarray(left, dtype=np.uint64)
        b = np.ascontiguousarray(right, dtype=np.uint64)
        out = np.empty(L, dtype=np.uint64)
        assert self.lib.orc_merge(field, width, _p64(a), _p64(b), _p64(out)) == 0
        return out

    def digest_bytes(self, field, digest):
        L = self.limbs(field)
        d = np.ascontiguousarray(digest, dtype=np.uint64)
        out = np.empty(8 * L, dtype=np.uint8)
        assert self.lib.orc_digest_bytes(field, _p64(d), _p8(out)) == 0
        return out.tobytes()

    def merkle_root(self, field, leaves, depth):
        L = self.limbs(field)
        lv = np.ascontiguousarray(leaves, dtype=np.uint64).reshape(-1, L)
        assert len(lv) == 1 << depth
        out = np.empty(L, dtype=np.uint64)
        assert self.lib.orc_merkle_root(field, _p64(lv), depth, _p64(out)) == 0
        return out

    # ---- batches (threaded; also the CPU baseline)
    def compress_batch(self, field, width, states, k=2, threads=1):
        L = self.limbs(field)
        s = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, width, L)
        out = np.empty((len(s), width // k, L), dtype=np.uint64)
        rc = self.lib.orc_compress_batch(field, width, k, _p64(s), _p64(out), len(s), threads)
        if rc:
            raise ValueError("orc_compress_batch rc=%d" % rc)
        return out

    def hash_bytes_batch(self, field, width, msgs, threads=1):
        L = self.limbs(field)
        m = np.ascontiguousarray(msgs, dtype=np.uint8)
        assert m.ndim == 2
        out = np.empty((m.shape[0], L), dtype=np.uint64)
        ptr = _p8(m) if m.size else _p8(np.zeros(1, dtype=np.uint8))
        assert self.lib.orc_hash_bytes_batch(field, width, ptr, m.shape[1], m.shape[0], _p64(out), threads) == 0
        return out

    def hash_field_batch(self, field, width, elems, threads=1):
        L = self.limbs(field)
        e = np.ascontiguousarray(elems, dtype=np.uint64)
        assert e.ndim == 3 and e.shape[2] == L
        out = np.empty((e.shape[0], L), dtype=np.uint64)
        ptr = _p64(e) if e.size else _p64(np.zeros(1, dtype=np.uint64))
        assert self.lib.orc_hash_field_batch(field, width, ptr, e.shape[1], e.shape[0], _p64(out), threads) == 0
        return out
