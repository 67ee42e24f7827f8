"""Pins both oracles (C: oracle/anemoi_oracle.c, Python: oracle/anemoi_ref.py) against every
known-answer vector of the reference's own unit tests (tests/golden/kats.json, text-extracted by
tools/extract_fixtures.py) and against each other on seeded random inputs.  CPU only."""
import random

import numpy as np
import pytest

from conftest import FIELD_IDS, INSTANCES, inst_key
from anemoi_ref import Instance


def ints(v):
    return [int(x) for x in v]


@pytest.mark.parametrize("field,width", INSTANCES)
def test_python_ref_reference_kats(kats, field, width):
    k, I = kats[inst_key(field, width)], Instance(field, width)
    for a, b in zip(k["sbox"]["in"], k["sbox"]["out"]):        # src/<f>/anemoi_*/mod.rs test_sbox
        st = ints(a)
        I.sbox_layer(st)
        assert st == ints(b)
    for a, b in zip(k["hash_field"]["in"], k["hash_field"]["out"]):   # hasher.rs test_anemoi_hash
        assert I.hash_field(ints(a)) == int(b)
    for a, b in zip(k["hash_bytes"]["in_hex"], k["hash_bytes"]["out"]):  # test_anemoi_hash_bytes
        assert I.hash(bytes.fromhex(a)) == int(b)
    for a, b in zip(k["jive"]["in"], k["jive"]["out"]):        # test_anemoi_jive
        assert I.compress(ints(a)) == ints(b)
        assert I.compress_k(ints(a), 2) == ints(b)
        if width == 2:
            assert I.merge(int(a[0]), int(a[1])) == int(b[0])
    if width == 4:
        for a, b in zip(k["jive_k4"]["in"], k["jive_k4"]["out"]):
            assert I.compress_k(ints(a), 4) == ints(b)


@pytest.mark.parametrize("field,width", INSTANCES)
def test_c_oracle_reference_kats(kats, oracle, field, width):
    k, fid = kats[inst_key(field, width)], FIELD_IDS.index(field)
    m = lambda v: oracle.ints_to_mont(fid, ints(v))
    back = lambda a: oracle.mont_to_ints(fid, a)
    for a, b in zip(k["sbox"]["in"], k["sbox"]["out"]):
        assert back(oracle.sbox_layer(fid, width, m(a))) == ints(b)
    for a, b in zip(k["hash_field"]["in"], k["hash_field"]["out"]):
        assert back(oracle.hash_field(fid, width, m(a))) == [int(b)]
    for a, b in zip(k["hash_bytes"]["in_hex"], k["hash_bytes"]["out"]):
        assert back(oracle.hash_bytes(fid, width, bytes.fromhex(a))) == [int(b)]
    for a, b in zip(k["jive"]["in"], k["jive"]["out"]):
        assert back(oracle.compress_k(fid, width, m(a), 2)) == ints(b)
        if width == 2:
            assert back(oracle.merge(fid, width, m(a)[0], m(a)[1])) == ints(b)
    if width == 4:
        for a, b in zip(k["jive_k4"]["in"], k["jive_k4"]["out"]):
            assert back(oracle.compress_k(fid, width, m(a), 4)) == ints(b)


@pytest.mark.parametrize("field,width", INSTANCES)
def test_c_oracle_vs_python_ref_random(oracle, field, width):
    """Differential check on what the reference leaves unpinned: random compress / permutation,
    hash(bytes) with partial last chunks and empty input, merge, digest bytes, Merkle roots."""
    I, fid = Instance(field, width), FIELD_IDS.index(field)
    rng = random.Random(0xA9E30100 + 16 * fid + width)
    for _ in range(3):
        st = [rng.randrange(I.p) for _ in range(width)]
        mont = oracle.ints_to_mont(fid, st)
        assert oracle.mont_to_ints(fid, mont) == st
        assert oracle.mont_to_ints(fid, oracle.permutation(fid, width, mont)) == I.permutation(list(st))
        assert oracle.mont_to_ints(fid, oracle.compress_k(fid, width, mont, 2)) == I.compress(st)
        assert oracle.mont_to_ints(fid, oracle.merge(fid, width, mont[0], mont[1])) == [I.merge(st[0], st[1])]
        assert oracle.digest_bytes(fid, mont[0]) == I.digest_to_bytes(st[0])
    for ln in [0, 1, I.chunk - 1, I.chunk, I.chunk + 1, 3 * I.chunk, 3 * I.chunk + 5, 200]:
        msg = bytes(rng.randrange(256) for _ in range(ln))
        assert oracle.mont_to_ints(fid, oracle.hash_bytes(fid, width, msg)) == [I.hash(msg)]
    for ne in [0, 1, 2, 3, 4, 7]:
        el = [rng.randrange(I.p) for _ in range(ne)]
        mont = oracle.ints_to_mont(fid, el) if ne else np.zeros((0, I.limbs), dtype=np.uint64)
        assert oracle.mont_to_ints(fid, oracle.hash_field(fid, width, mont)) == [I.hash_field(el)]
    with pytest.raises(ValueError):
        oracle.compress_k(fid, width, oracle.ints_to_mont(fid, [1] * width), 3)


@pytest.mark.parametrize("field", ["jubjub", "bls12_381"])
def test_merkle_root_small(oracle, field):
    I, fid = Instance(field, 2), FIELD_IDS.index(field)
    rng = random.Random(7)
    leaves = [rng.randrange(I.p) for _ in range(8)]
    root = oracle.merkle_root(fid, oracle.ints_to_mont(fid, leaves), 3)
    assert oracle.mont_to_ints(fid, root) == [I.merkle_root(leaves)]


def test_batch_threads_match_single(oracle):
    fid, width = FIELD_IDS.index("vesta"), 2
    I = Instance("vesta", 2)
    rng = random.Random(11)
    st = oracle.ints_to_mont(fid, [rng.randrange(I.p) for _ in range(2 * 32)]).reshape(32, 2, 4)
    a = oracle.compress_batch(fid, width, st, threads=1)
    b = oracle.compress_batch(fid, width, st, threads=4)
    assert (a == b).all()
    assert (a[5] == oracle.compress_k(fid, width, st[5], 2)).all()


def test_c_oracle_matches_minted_goldens(oracle):
    """tests/golden/extra.json is minted from the Python restatement (tools/mint_goldens.py); the C
    oracle must reproduce it (cross-check of the two restatements on the unpinned cases)."""
    import json, os
    from conftest import ROOT
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "extra.json")))
    for v in g["hash_bytes"]:
        fid = FIELD_IDS.index(v["field"])
        assert oracle.mont_to_ints(fid, oracle.hash_bytes(fid, v["width"], bytes.fromhex(v["msg_hex"]))) == [int(v["digest"])]
    for v in g["compress"]:
        fid = FIELD_IDS.index(v["field"])
        got = oracle.compress_k(fid, v["width"], oracle.ints_to_mont(fid, ints(v["in"])), 2)
        assert oracle.mont_to_ints(fid, got) == ints(v["out"])
    for v in g["merkle"]:
        fid = FIELD_IDS.index(v["field"])
        got = oracle.merkle_root(fid, oracle.ints_to_mont(fid, ints(v["leaves"])), v["depth"])
        assert oracle.mont_to_ints(fid, got) == [int(v["root"])]
