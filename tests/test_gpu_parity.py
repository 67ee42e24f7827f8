"""GPU parity: the HIP path (through the C-ABI, via anemoi_amd) against
  (1) every known-answer vector of the reference's own tests (tests/golden/kats.json),
  (2) the pinned C oracle on seeded random batches incl. ragged/empty/partial-chunk edge cases,
  (3) size-independent properties at BASELINE.json's full sizes.
Bit-exact everywhere (integer arithmetic).  Run on the GPU box: pytest -m gpu.
"""
import os
import random

import numpy as np
import pytest

from conftest import FIELD_IDS, INSTANCES, inst_key, knobs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import anemoi_amd
    assert anemoi_amd.device_count() >= 1
    return anemoi_amd


def ints(v):
    return [int(x) for x in v]


def rand_elems(oracle, fid, modulus, count, seed):
    rng = random.Random(seed)
    return oracle.ints_to_mont(fid, [rng.randrange(modulus) for _ in range(count)])


# small batches take the cooperative latency kernels by default -- the two-row fold kernels (coop2d.h: two 2-1 items or
# one 4-3 state per wavefront); "row" switches the fold kernels off (everything on the row-cooperative scan
# kernels), "lane" forces the lane-private throughput kernels.  (The one-item-per-wavefront kernels -- four-row fold /
# scan, the recorded negatives -- exist in `make AB=1` libraries only; test_cooperative_and_lane_private_paths_agree and
# the fuzz force them when ANEMOI_MI355X_LIB points at one: the product ignores ANEMOI_COOP_MAX.)
LATENCY_KERNELS = {"default": {}, "row": {"coop2d_max": 0, "coop2d43_max": 0},
                   "lane": {"coop2d_max": 0, "coop4_max": 0, "coop43_max": 0, "coop2d43_max": 0,
                            "coop_sponge_max": 0, "coop_climb_max": 0}}


# ---------------------------------------------------------------- (1) the reference's own KATs

@pytest.mark.parametrize("kernels", ["default", "row", "lane"])
@pytest.mark.parametrize("field,width", INSTANCES)
def test_reference_kats(A, kats, field, width, kernels):
    with knobs(**LATENCY_KERNELS[kernels]):
        check_reference_kats(A, kats, field, width)


def check_reference_kats(A, kats, field, width):
    k, inst = kats[inst_key(field, width)], A.Anemoi(field, width)
    enc, dec = inst.encode, inst.decode

    # test_sbox (src/<f>/anemoi_x/mod.rs:68): sbox_layer on 10 states
    sin = np.stack([enc(ints(a)) for a in k["sbox"]["in"]])
    got = inst.sbox_layer_batch(sin)
    for g, b in zip(got, k["sbox"]["out"]):
        assert dec(g) == ints(b)

    # test_anemoi_hash (hasher.rs:123): hash_field on 10 inputs of different lengths
    for a, b in zip(k["hash_field"]["in"], k["hash_field"]["out"]):
        assert dec(inst.hash_field(enc(ints(a)))) == [int(b)]

    # test_anemoi_hash_bytes (hasher.rs:202)
    for a, b in zip(k["hash_bytes"]["in_hex"], k["hash_bytes"]["out"]):
        assert dec(inst.hash(bytes.fromhex(a))) == [int(b)]

    # test_anemoi_jive (hasher.rs:231): compress, compress_k(.,2), merge (2-1), compress_k(.,4) (4-3)
    for a, b in zip(k["jive"]["in"], k["jive"]["out"]):
        e = enc(ints(a))
        assert dec(inst.compress(e)) == ints(b)
        assert dec(inst.compress_k(e, 2)) == ints(b)
        if width == 2:
            assert dec(inst.merge(e)) == ints(b)
    if width == 4:
        for a, b in zip(k["jive_k4"]["in"], k["jive_k4"]["out"]):
            assert dec(inst.compress_k(enc(ints(a)), 4)) == ints(b)
    # the reference's assert!s
    with pytest.raises(A.AnemoiError):
        inst.compress_k(enc([1] * width), 3)
    if width == 2:
        with pytest.raises(A.AnemoiError):
            inst.compress_k(enc([1, 1]), 4)


# ---------------------------------------------------------------- (2) differential vs the oracle

@pytest.mark.parametrize("kernels", ["default", "row", "lane"])
@pytest.mark.parametrize("field,width", INSTANCES)
def test_permutation_and_jive_vs_oracle(A, oracle, params, field, width, kernels):
    """batches of 1 .. 130 states: by default the row-cooperative kernels (k_jive2_coop / k_jive4_coop /
    k_permutation_coop), with "lane" the lane-private / lane-pair throughput kernels"""
    with knobs(**LATENCY_KERNELS[kernels]):
        check_permutation_and_jive_vs_oracle(A, oracle, params, field, width)


def check_permutation_and_jive_vs_oracle(A, oracle, params, field, width):
    fid, p = FIELD_IDS.index(field), int(params[field]["modulus"])
    inst, L = A.Anemoi(field, width), params[field]["u64_limbs"]
    # ragged batch sizes around the 64-lane workgroup
    for n in (1, 63, 64, 65, 130):
        st = rand_elems(oracle, fid, p, n * width, 1000 * fid + 10 * width + n).reshape(n, width, L)
        # edge states: all-zero, all p-1
        st[0] = 0
        if n > 1:
            st[1] = oracle.ints_to_mont(fid, [p - 1] * width)
        got_p = inst.permutation_batch(st)
        got_c = inst.compress_batch(st)
        exp_c = oracle.compress_batch(fid, width, st, k=2, threads=8)
        assert (got_c == exp_c).all()
        for i in (0, n // 2, n - 1):
            assert (got_p[i] == oracle.permutation(fid, width, st[i])).all()
        if width == 4:
            assert (inst.compress_k_batch(st, 4) == oracle.compress_batch(fid, width, st, k=4, threads=8)).all()
        else:
            assert (inst.merge_batch(st) == exp_c[:, 0]).all()
    # 4-3 merge reproduces the reference's behaviour (digests[0] in both rate cells)
    if width == 4:
        pr = rand_elems(oracle, fid, p, 2 * 5, 77).reshape(5, 2, L)
        got = inst.merge_batch(pr)
        for i in range(5):
            assert (got[i] == oracle.merge(fid, width, pr[i, 0], pr[i, 1])).all()


@pytest.mark.parametrize("kernels", ["default", "row", "lane"])
@pytest.mark.parametrize("field,width", INSTANCES)
def test_sponge_vs_oracle(A, oracle, params, field, width, kernels):
    """small batches: by default the row-cooperative sponge (k_sponge_coop), with "lane" the lane-private one"""
    with knobs(**LATENCY_KERNELS[kernels]):
        check_sponge_vs_oracle(A, oracle, params, field, width)


def check_sponge_vs_oracle(A, oracle, params, field, width):
    fid, p = FIELD_IDS.index(field), int(params[field]["modulus"])
    inst, L, ch = A.Anemoi(field, width), params[field]["u64_limbs"], params[field]["byte_chunk"]
    rng = np.random.default_rng(fid * 10 + width)
    # byte messages: empty, 1 byte, around one chunk, around the rate boundary, partial last chunk
    for ln in (0, 1, ch - 1, ch, ch + 1, 2 * ch, 3 * ch - 1, 3 * ch, 3 * ch + 1, 6 * ch + 5, 200):
        n = 5 if ln < 100 else 3
        msgs = rng.integers(0, 256, size=(n, ln), dtype=np.uint8)
        if ln:
            msgs[0] = 0            # all-zero message still gets the 0x01 pad
            msgs[-1] = 255
        got = inst.hash_batch(msgs)
        exp = oracle.hash_bytes_batch(fid, width, msgs, threads=4)
        assert (got == exp).all(), (field, width, ln)
    # element messages of every length class mod RATE, incl. empty
    for ne in (0, 1, 2, 3, 4, 5, 6, 7):
        n = 4
        el = rand_elems(oracle, fid, p, n * ne, 31 * ne + fid).reshape(n, ne, L) if ne else np.zeros((n, 0, L), np.uint64)
        got = inst.hash_field_batch(el)
        exp = oracle.hash_field_batch(fid, width, el, threads=4)
        assert (got == exp).all(), (field, width, ne)
    # digest bytes (digest.rs:42-46)
    d = rand_elems(oracle, fid, p, 1, 5)[0]
    assert inst.digest_to_bytes(d) == oracle.digest_bytes(fid, d)


@pytest.mark.parametrize("field,width", INSTANCES)
def test_long_sponge_bound_stress(A, oracle, params, field, width):
    """Many chained permutations per lane (the unsaturated-limb arithmetic keeps values only loosely
    bounded between reductions): 3 000-byte messages incl. all-0xFF and all-zero ones, and element
    messages made of p-1, against the oracle."""
    fid, p, L = FIELD_IDS.index(field), int(params[field]["modulus"]), params[field]["u64_limbs"]
    inst = A.Anemoi(field, width)
    rng = np.random.default_rng(99 + fid)
    msgs = rng.integers(0, 256, size=(6, 3000), dtype=np.uint8)
    msgs[0] = 0xFF
    msgs[1] = 0
    el = np.broadcast_to(oracle.ints_to_mont(fid, [p - 1]), (2, 40, L)).copy()
    el[1, ::2] = oracle.ints_to_mont(fid, [1])
    for kernels in ("default", "lane"):   # the row-cooperative and the lane-private sponge
        with knobs(**LATENCY_KERNELS[kernels]):
            assert (inst.hash_batch(msgs) == oracle.hash_bytes_batch(fid, width, msgs, threads=6)).all(), kernels
            assert (inst.hash_field_batch(el) == oracle.hash_field_batch(fid, width, el, threads=2)).all(), kernels


@pytest.mark.parametrize("field", FIELD_IDS)
def test_montgomery_conversion(A, oracle, params, field):
    fid, p, L = FIELD_IDS.index(field), int(params[field]["modulus"]), params[field]["u64_limbs"]
    rng = random.Random(fid)
    vals = [0, 1, p - 1, p - 2] + [rng.randrange(p) for _ in range(96)]
    canon = A.ints_to_limbs(vals, L)
    mont = A.to_montgomery(field, canon)
    assert (mont == oracle.ints_to_mont(fid, vals)).all()
    assert (A.from_montgomery(field, mont) == canon).all()


# ---------------------------------------------------------------- the input contract: ANY 64 L-bit pattern is taken mod p

def unreduced(p, L, count, seed):
    """`count` raw 64 L-bit integers, most of them NOT below p: p, p + 1, 2 p - 1 ... 2^(64 L) - 1, multiples of p, random
    values of the whole range (on ed_on_bls12_377 up to 13.7 p, on bls12_377 up to 152 p), a few reduced ones between"""
    top = 1 << (64 * L)
    rng = random.Random(seed)
    fixed = [p, p + 1, top - 1, top - 2, min(2 * p - 1, top - 1), min(2 * p, top - 1), (top - 1) // p * p, (top - 1) // p * p - 1,
             0, 1, p - 1, top >> 1]
    vals = fixed + [rng.randrange(p, top) for _ in range(count)]
    for i in range(len(fixed), len(vals), 7):
        vals[i] = rng.randrange(p)
    return vals[:count]


@pytest.mark.parametrize("kernels", ["default", "row", "lane"])
@pytest.mark.parametrize("field,width", INSTANCES)
def test_unreduced_abi_inputs_are_taken_mod_p(A, oracle, params, field, width, kernels):
    """Round 5's review, item 4.  arkworks keeps a `Felt` below p (src/<field>/mod.rs:1-3; `from_le_bytes_mod_order`,
    anemoi_2_1/hasher.rs:57), but a C-ABI sees whatever words the caller's buffer holds.  The header now says: any 64 L-bit
    pattern X is taken as X mod p -- proven by the bounds walk, which starts every ABI input at 2^(64 L) - 1
    (tests/cpp/bounds_walk/bounds_walk.cpp: LayoutInfo::abi_max; csrc/BOUNDS.md rows `from_abi`), and checked here on
    the device: every entry point that reads ABI elements, under the three kernel routings, fed with p, p + 1,
    2^(64 L) - 1, multiples of p and random patterns of the whole range, must return EXACTLY what the oracle returns
    for the reduced values."""
    fid, p, L = FIELD_IDS.index(field), int(params[field]["modulus"]), params[field]["u64_limbs"]
    inst = A.Anemoi(field, width)
    with knobs(**LATENCY_KERNELS[kernels]):
        for n in (1, 3, 70):
            raw_i = unreduced(p, L, n * width, 31 * fid + width + n)
            raw = A.ints_to_limbs(raw_i, L).reshape(n, width, L)
            red = A.ints_to_limbs([v % p for v in raw_i], L).reshape(n, width, L)
            assert n < 3 or (raw != red).any()
            assert (inst.compress_batch(raw) == oracle.compress_batch(fid, width, red, k=2, threads=8)).all(), n
            gp, gs = inst.permutation_batch(raw), inst.sbox_layer_batch(raw)
            for i in (0, n // 2, n - 1):
                assert (gp[i] == oracle.permutation(fid, width, red[i])).all(), (n, i)
                assert (gs[i] == oracle.sbox_layer(fid, width, red[i])).all(), (n, i)
            if width == 4:
                assert (inst.compress_k_batch(raw, 4) == oracle.compress_batch(fid, width, red, k=4, threads=8)).all(), n
            else:
                assert (inst.merge_batch(raw) == oracle.compress_batch(fid, width, red, k=2, threads=8)[:, 0]).all(), n
            # hash_field: equal-length messages of 1 .. 7 elements (the rate block and its padding), and a ragged batch
            for epm in (1, 2, 3, 4, 7):
                e_i = unreduced(p, L, n * epm, 5 * fid + epm + n)
                e_raw = A.ints_to_limbs(e_i, L).reshape(n, epm, L)
                e_red = A.ints_to_limbs([v % p for v in e_i], L).reshape(n, epm, L)
                assert (inst.hash_field_batch(e_raw) == oracle.hash_field_batch(fid, width, e_red, threads=8)).all(), (n, epm)
            counts = [(3 * i + n) % 6 for i in range(n)]
            e_i = unreduced(p, L, sum(counts) + 1, 77 + fid)
            e_raw, e_red = A.ints_to_limbs(e_i, L), A.ints_to_limbs([v % p for v in e_i], L)
            at, m_raw, m_red = 0, [], []
            for c in counts:
                m_raw.append(e_raw[at:at + c]), m_red.append(e_red[at:at + c])
                at += c
            got = inst.hash_field_ragged(m_raw)
            for i in range(n):
                assert (got[i] == oracle.hash_field(fid, width, m_red[i])).all(), (n, i, counts[i])
    # conversions (always on the 32-bit-limb form): x R and x / R of the reduced value, canonical
    v_i = unreduced(p, L, 100, 900 + fid)
    v_raw, v_red = A.ints_to_limbs(v_i, L), A.ints_to_limbs([v % p for v in v_i], L)
    assert (A.to_montgomery(field, v_raw) == A.to_montgomery(field, v_red)).all()
    assert (A.from_montgomery(field, v_raw) == A.from_montgomery(field, v_red)).all()
    assert (A.from_montgomery(field, v_raw) == A.ints_to_limbs(oracle.mont_to_ints(fid, v_red), L)).all()
    if width == 2:
        # Merkle: unreduced leaves (the retained level 0 is the caller's own bytes, every level above it is canonical), and
        # path verification with unreduced leaves AND unreduced siblings against the canonical root (the `root` argument
        # itself is compared as bytes with a canonical value: it must be reduced, as every root the library returns is)
        with knobs(**LATENCY_KERNELS[kernels]):
            depth = 5
            l_i = unreduced(p, L, 1 << depth, 4000 + fid)
            l_raw, l_red = A.ints_to_limbs(l_i, L), A.ints_to_limbs([v % p for v in l_i], L)
            root = oracle.merkle_root(fid, l_red, depth)
            assert (inst.merkle_root(l_raw, depth) == root).all()
            lv_raw, lv_red = inst.merkle_tree(l_raw, depth), [a.copy() for a in inst.merkle_tree(l_red, depth)]
            assert (lv_raw[0] == l_raw).all() and all((a == b).all() for a, b in zip(lv_raw[1:], lv_red[1:]))
            idx = [0, 5, 17, 31]
            paths = np.stack([inst.merkle_path(lv_red, depth, i) for i in idx])
            paths[:, 0] = l_raw[[i ^ 1 for i in idx]]          # the bottom sibling as the caller's unreduced bytes
            top_room = ((1 << (64 * L)) - 1) // p               # ... and every other sibling shifted by a multiple of p
            lifted = [[v + (k % top_room) * p for k, v in enumerate(A.limbs_to_ints(pth[1:]), 1)] for pth in paths]
            paths[:, 1:] = np.stack([A.ints_to_limbs(row, L) for row in lifted])
            assert inst.merkle_verify_batch(l_raw[idx], idx, paths, depth, root).all()
            assert not inst.merkle_verify_batch(l_raw[idx], [i ^ 2 for i in idx], paths, depth, root).any()


@pytest.mark.parametrize("field", ["jubjub", "bls12_381", "vesta"])
def test_merkle_vs_oracle(A, oracle, params, field):
    fid, p, L = FIELD_IDS.index(field), int(params[field]["modulus"]), params[field]["u64_limbs"]
    inst = A.Anemoi(field, 2)
    for depth in (0, 1, 2, 5, 8):
        leaves = rand_elems(oracle, fid, p, 1 << depth, 900 + depth)
        assert (inst.merkle_root(leaves, depth) == oracle.merkle_root(fid, leaves, depth)).all()
    # sharded driver (ALL_DEVICES) must agree with the single-device one
    leaves = rand_elems(oracle, fid, p, 1 << 7, 4242)
    assert (A.Anemoi(field, 2, device=A.ALL_DEVICES).merkle_root(leaves, 7) == inst.merkle_root(leaves, 7)).all()


@pytest.mark.parametrize("field", ["jubjub", "bls12_381", "ed_on_bls12_377"])
def test_merkle_tree_paths_verify(A, oracle, params, field):
    """Retained-level tree, authentication paths and batched verification vs the Python restatement."""
    from anemoi_ref import Instance
    I, fid = Instance(field, 2), FIELD_IDS.index(field)
    inst, depth = A.Anemoi(field, 2), 5
    rng = random.Random(31 + fid)
    leaves_i = [rng.randrange(I.p) for _ in range(1 << depth)]
    levels_i = I.merkle_levels(leaves_i)
    leaves = oracle.ints_to_mont(fid, leaves_i)
    levels = inst.merkle_tree(leaves, depth)
    assert [len(l) for l in levels] == [32, 16, 8, 4, 2, 1]
    for got, exp in zip(levels, levels_i):
        assert inst.decode(got) == exp
    idx = [0, 1, 13, 31, 18]
    paths = np.stack([inst.merkle_path(levels, depth, i) for i in idx])
    for i, pth in zip(idx, paths):
        assert inst.decode(pth) == I.merkle_path(levels_i, i)
        assert I.merkle_climb(leaves_i[i], i, I.merkle_path(levels_i, i)) == levels_i[-1][0]
    ok = inst.merkle_verify_batch(leaves[idx], idx, paths, depth, levels[-1][0])
    assert ok.all()
    # tampering: wrong index, wrong leaf, wrong sibling must all be rejected
    bad_idx = inst.merkle_verify_batch(leaves[idx], [1, 0, 12, 30, 19], paths, depth, levels[-1][0])
    assert not bad_idx.any()
    bad_paths = paths.copy()
    bad_paths[:, 2, 0] ^= np.uint64(1)
    assert not inst.merkle_verify_batch(leaves[idx], idx, bad_paths, depth, levels[-1][0]).any()
    assert not inst.merkle_verify_batch(leaves[[1, 0, 14, 30, 17]], idx, paths, depth, levels[-1][0]).any()
    # depth 0: the leaf is the root
    assert inst.merkle_verify_batch(leaves[:1], [0], np.zeros((1, 0, inst.limbs), np.uint64), 0, leaves[0]).all()


@pytest.mark.parametrize("field", ["bn_254", "bls12_377"])
def test_merkle_arity4(A, oracle, params, field):
    from anemoi_ref import Instance
    I, fid = Instance(field, 4), FIELD_IDS.index(field)
    inst = A.Anemoi(field, 4)
    rng = random.Random(77 + fid)
    for depth4 in (0, 1, 3):
        leaves_i = [rng.randrange(I.p) for _ in range(4 ** depth4)]
        got = inst.merkle_root_arity4(oracle.ints_to_mont(fid, leaves_i), depth4)
        assert inst.decode(got) == [I.merkle_root_arity4(leaves_i)]


@pytest.mark.parametrize("field", ["bn_254", "bls12_381", "pallas"])
def test_merkle_arity4_tree_paths_verify(A, oracle, params, field):
    """arity-4 tree with retained levels, authentication paths and batched verification vs the oracle"""
    from anemoi_ref import Instance
    I, fid = Instance(field, 4), FIELD_IDS.index(field)
    inst = A.Anemoi(field, 4)
    rng = random.Random(401 + fid)
    for depth4 in (0, 1, 3):
        n = 4 ** depth4
        leaves_i = [rng.randrange(I.p) for _ in range(n)]
        ref_levels = I.merkle_levels_arity4(leaves_i)
        leaves = oracle.ints_to_mont(fid, leaves_i)
        levels = inst.merkle_tree_arity4(leaves, depth4)
        assert len(levels) == depth4 + 1
        for got, exp in zip(levels, ref_levels):
            assert inst.decode(got) == exp
        assert (levels[-1][0] == inst.merkle_root_arity4(leaves, depth4)).all()
        idx = sorted({0, n - 1, n // 3, (2 * n) // 3, rng.randrange(n)})
        paths = np.stack([inst.merkle_path_arity4(levels, depth4, i) for i in idx]) if depth4 else \
            np.zeros((len(idx), 0, inst.limbs), dtype=np.uint64)
        for k, i in enumerate(idx):
            exp_path = I.merkle_path_arity4(ref_levels, i)
            assert inst.decode(paths[k]) == exp_path
            assert I.merkle_climb_arity4(leaves_i[i], i, exp_path) == ref_levels[-1][0]
        root = levels[-1][0]
        sel = leaves[idx]
        assert inst.merkle_verify_arity4_batch(sel, idx, paths, depth4, root).all()
        if depth4:
            bad = sel.copy()
            bad[0, 0] ^= np.uint64(1)                      # tampered leaf
            assert not inst.merkle_verify_arity4_batch(bad, idx, paths, depth4, root)[0]
            wrong_idx = list(idx)
            wrong_idx[-1] ^= 1                             # right leaf, wrong slot
            assert not inst.merkle_verify_arity4_batch(sel, wrong_idx, paths, depth4, root)[-1]
            badp = paths.copy()
            badp[1, -1, 0] ^= np.uint64(2)                 # tampered top sibling
            res = inst.merkle_verify_arity4_batch(sel, idx, badp, depth4, root)
            assert not res[1] and res[0]
    with pytest.raises(A.AnemoiError):
        inst.merkle_path_arity4(levels, 3, 64)             # index out of range
    with pytest.raises(A.AnemoiError):
        A.Anemoi(field, 4).merkle_tree_arity4(leaves[:5], 1)


def test_golden_extra_vectors(A, oracle):
    """Vectors minted by tools/mint_goldens.py from the Python big-int restatement for the cases the
    reference's tests leave unpinned (partial chunk, empty input, 10 KB messages, Merkle roots)."""
    import json, os
    from conftest import ROOT
    path = os.path.join(ROOT, "tests", "golden", "extra.json")
    g = json.load(open(path))
    for v in g["hash_bytes"]:
        inst = A.Anemoi(v["field"], v["width"])
        assert inst.decode(inst.hash(bytes.fromhex(v["msg_hex"]))) == [int(v["digest"])]
    for v in g["compress"]:
        inst = A.Anemoi(v["field"], v["width"])
        assert inst.decode(inst.compress(inst.encode(ints(v["in"])))) == ints(v["out"])
    for v in g["merkle"]:
        inst = A.Anemoi(v["field"], 2)
        assert inst.decode(inst.merkle_root(inst.encode(ints(v["leaves"])), v["depth"])) == [int(v["root"])]


# ---------------------------------------------------------------- (3) full-size configs

def test_cfg1_vesta_1024(A, oracle, params):
    """BASELINE config 1: Anemoi-2-1 over Vesta, 1024 Jive compressions: every output vs the oracle."""
    fid, p = FIELD_IDS.index("vesta"), int(params["vesta"]["modulus"])
    st = rand_elems(oracle, fid, p, 2048, 0xA9E30101).reshape(1024, 2, 4)
    assert (A.Anemoi("vesta", 2).compress_batch(st) == oracle.compress_batch(fid, 2, st, threads=8)).all()


def test_cfg2_bls12_381_2pow20(A, oracle, params):
    """BASELINE config 2: 2^20 BLS12-381 Anemoi-2-1 compressions drawn from 4 096 distinct states, every one checked against the oracle, +
    size-independent properties: batch == items run alone, shuffle-equivariance, ALL_DEVICES == 1 GPU."""
    fid, p, n = 0, int(params["bls12_381"]["modulus"]), 1 << 20
    rng = np.random.default_rng(0xA9E30102)
    base = rand_elems(oracle, fid, p, 2 * 4096, 0xA9E30102).reshape(4096, 2, 6)
    idx = rng.integers(0, 4096, size=n)
    st = base[idx]                                   # 2^20 states drawn from 4096 distinct ones
    inst = A.Anemoi("bls12_381", 2)
    out = inst.compress_batch(st)
    assert out.shape == (n, 1, 6)
    ref = inst.compress_batch(base)
    assert (ref == oracle.compress_batch(fid, 2, base, threads=8)).all()      # all 4 096 distinct states
    # equal inputs -> equal outputs, everywhere in the batch (catches any index-dependent corruption)
    assert (out == ref[idx]).all()
    # sharding over all visible GPUs gives the same bytes
    assert (A.Anemoi("bls12_381", 2, device=A.ALL_DEVICES).compress_batch(st[: 1 << 16]) == out[: 1 << 16]).all()


def test_cfg3_bn254_sponge_10k(A, oracle):
    """BASELINE config 3 shape (Anemoi-4-3 over BN-254, 10 240-byte messages; 10240 % 31 = 10 so the
    partial-chunk padding branch runs): 256 messages on the GPU, 24 of them against the oracle, the
    rest by duplicate-consistency."""
    fid = FIELD_IDS.index("bn_254")
    rng = np.random.default_rng(0xA9E30103)
    base = rng.integers(0, 256, size=(24, 10240), dtype=np.uint8)
    idx = rng.integers(0, 24, size=256)
    got = A.Anemoi("bn_254", 4).hash_batch(base[idx])
    exp = oracle.hash_bytes_batch(fid, 4, base, threads=8)
    assert (got == exp[idx]).all()


def test_cfg3_full_size_2pow16_messages(A, oracle):
    """BASELINE config 3 at its full size: 2^16 messages x 10 240 bytes (640 MiB).  Messages are drawn
    from 16 distinct ones, so every digest is known from the oracle: checks all 65 536 lanes."""
    fid = FIELD_IDS.index("bn_254")
    rng = np.random.default_rng(0xA9E30133)
    base = rng.integers(0, 256, size=(16, 10240), dtype=np.uint8)
    idx = rng.integers(0, 16, size=1 << 16)
    got = A.Anemoi("bn_254", 4).hash_batch(base[idx])
    exp = oracle.hash_bytes_batch(fid, 4, base, threads=8)
    assert (got == exp[idx]).all()


def test_cfg4_per_gpu_share_2pow21(A, oracle, params):
    """BASELINE config 4's per-GPU share (2^24 / 8 = 2^21 BLS12-381 compressions): equal inputs give
    equal outputs across the whole batch, the distinct ones match the oracle."""
    fid, p = 0, int(params["bls12_381"]["modulus"])
    rng = np.random.default_rng(0xA9E30104)
    base = rand_elems(oracle, fid, p, 2 * 256, 0xA9E30104).reshape(256, 2, 6)
    idx = rng.integers(0, 256, size=1 << 21)
    out = A.Anemoi("bls12_381", 2).compress_batch(base[idx])
    exp = oracle.compress_batch(fid, 2, base, threads=8)
    assert (out == exp[idx]).all()


def test_cfg5_per_gpu_subtree_depth21(A, oracle, params):
    """BASELINE config 5's per-GPU subtree (2^21 Jubjub leaves): root == merge(root(left half),
    root(right half)) (a size-independent property; depth 14 below is checked against the oracle) and
    the retained-level tree agrees with the root driver."""
    fid, p = FIELD_IDS.index("jubjub"), int(params["jubjub"]["modulus"])
    inst = A.Anemoi("jubjub", 2)
    rng = np.random.default_rng(0xA9E30155)
    leaves = rng.integers(0, 1 << 62, size=(1 << 21, 4), dtype=np.uint64)   # limbs < 2^62 => element < p
    root = inst.merkle_root(leaves, 21)
    l, r = inst.merkle_root(leaves[: 1 << 20], 20), inst.merkle_root(leaves[1 << 20:], 20)
    assert (inst.merge(np.stack([l, r])) == root).all()
    levels = inst.merkle_tree(leaves[: 1 << 16], 16)
    assert (levels[-1][0] == inst.merkle_root(leaves[: 1 << 16], 16)).all()
    assert (levels[1][:64] == oracle.compress_batch(fid, 2, leaves[:128].reshape(64, 2, 4), threads=8)[:, 0]).all()


def test_cfg5_jubjub_merkle_depth14(A, oracle, params):
    """BASELINE config 5 shape at depth 14 (2^14 leaves): root == merge of the two half-tree roots
    (recursively, a size-independent property) and == the oracle's root."""
    fid, p = FIELD_IDS.index("jubjub"), int(params["jubjub"]["modulus"])
    inst = A.Anemoi("jubjub", 2)
    leaves = rand_elems(oracle, fid, p, 1 << 14, 0xA9E30105)
    root = inst.merkle_root(leaves, 14)
    l, r = inst.merkle_root(leaves[: 1 << 13], 13), inst.merkle_root(leaves[1 << 13:], 13)
    assert (inst.merge(np.stack([l, r])) == root).all()
    assert (root == oracle.merkle_root(fid, leaves, 14)).all()


def test_cooperative_and_lane_private_paths_agree(oracle, params):
    """Jive 2-1 has three kernels: wave-cooperative with one item per wavefront (small batches), row-cooperative
    with four items per wavefront (one per 16-lane DPP row; medium batches) and lane-private (one item per lane);
    Jive 4-3 has the row-cooperative form with two states per wavefront and the lane-pair kernel.
    ANEMOI_COOP_MAX / ANEMOI_COOP4_MAX / ANEMOI_COOP43_MAX force each for every size; all must match the oracle bit
    for bit on all 7 fields, ragged sizes (partly filled wavefronts and rows) and edge states."""
    import subprocess, sys, os
    from conftest import ROOT
    code = r'''
import sys, os, random
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
import orc, anemoi_amd as A
oracle = orc.Oracle()
import json
params = json.load(open(os.path.join(ROOT, "tests", "golden", "params.json")))
for fid, field in enumerate(A.FIELD_IDS):
    p, L = int(params[field]["modulus"]), params[field]["u64_limbs"]
    rng = random.Random(fid)
    for n in (1, 2, 3, 7, 65, 300):
        st = oracle.ints_to_mont(fid, [rng.randrange(p) for _ in range(2 * n)]).reshape(n, 2, L)
        st[0] = 0
        if n > 1:
            st[1] = oracle.ints_to_mont(fid, [p - 1, p - 1])
        got = A.Anemoi(field, 2).compress_batch(st)
        assert (got == oracle.compress_batch(fid, 2, st, threads=8)).all(), (field, n)
    leaves = oracle.ints_to_mont(fid, [rng.randrange(p) for _ in range(64)])
    assert (A.Anemoi(field, 2).merkle_root(leaves, 6) == oracle.merkle_root(fid, leaves, 6)).all()
    # the other Anemoi-2-1 latency kernels under the same forcing: permutation, sponge (bytes / elements), path climb
    inst2 = A.Anemoi(field, 2)
    st = oracle.ints_to_mont(fid, [rng.randrange(p) for _ in range(2 * 5)]).reshape(5, 2, L)
    pg = inst2.permutation_batch(st)
    assert all((pg[i] == oracle.permutation(fid, 2, st[i])).all() for i in range(5)), field
    for ln in (0, 1, inst2.chunk - 1, inst2.chunk, inst2.chunk + 1, 100):
        msgs = np.frombuffer(rng.randbytes(3 * ln), dtype=np.uint8).reshape(3, ln) if ln else np.zeros((3, 0), dtype=np.uint8)
        assert (inst2.hash_batch(msgs) == oracle.hash_bytes_batch(fid, 2, msgs, threads=1)).all(), (field, ln)
    for ne in (0, 1, 2, 5):
        el = oracle.ints_to_mont(fid, [rng.randrange(p) for _ in range(3 * ne)]).reshape(3, ne, L)
        assert (inst2.hash_field_batch(el) == oracle.hash_field_batch(fid, 2, el, threads=1)).all(), (field, ne)
    tree = inst2.merkle_tree(leaves, 6)
    idx = [0, 1, 37, 63]
    paths = np.stack([inst2.merkle_path(tree, 6, i) for i in idx])
    okv = inst2.merkle_verify_batch(leaves[idx], np.array(idx, dtype=np.uint64), paths, 6, tree[-1][0])
    assert okv.all(), field
    bad = leaves[idx].copy(); bad[2, 0] ^= np.uint64(1)
    okv = inst2.merkle_verify_batch(bad, np.array(idx, dtype=np.uint64), paths, 6, tree[-1][0])
    assert list(okv) == [True, True, False, True], field
    # Anemoi-4-3: the row-cooperative kernel (two states per wavefront) against the lane-pair kernel's oracle, k = 2 and 4
    for n in (1, 2, 3, 33, 130):
        st = oracle.ints_to_mont(fid, [rng.randrange(p) for _ in range(4 * n)]).reshape(n, 4, L)
        st[0] = 0
        if n > 1:
            st[1] = oracle.ints_to_mont(fid, [p - 1] * 4)
        for k in (2, 4):
            got = A.Anemoi(field, 4).compress_k_batch(st, k)
            assert (got == oracle.compress_batch(fid, 4, st, k=k, threads=8)).all(), (field, n, k)
    # the other Anemoi-4-3 latency kernels under the same forcing: permutation, sponge (bytes / elements)
    inst4 = A.Anemoi(field, 4)
    st = oracle.ints_to_mont(fid, [rng.randrange(p) for _ in range(4 * 3)]).reshape(3, 4, L)
    pg = inst4.permutation_batch(st)
    assert all((pg[i] == oracle.permutation(fid, 4, st[i])).all() for i in range(3)), field
    for ln in (0, 1, inst4.chunk, 3 * inst4.chunk - 1, 3 * inst4.chunk, 3 * inst4.chunk + 1, 200):
        msgs = np.frombuffer(rng.randbytes(3 * ln), dtype=np.uint8).reshape(3, ln) if ln else np.zeros((3, 0), dtype=np.uint8)
        assert (inst4.hash_batch(msgs) == oracle.hash_bytes_batch(fid, 4, msgs, threads=1)).all(), (field, ln)
    for ne in (0, 1, 3, 4, 7):
        el = oracle.ints_to_mont(fid, [rng.randrange(p) for _ in range(3 * ne)]).reshape(3, ne, L)
        assert (inst4.hash_field_batch(el) == oracle.hash_field_batch(fid, 4, el, threads=1)).all(), (field, ne)
print("ok")
'''.replace("ROOT", repr(ROOT))
    # (lane-private; [laboratory build: one-per-wavefront 2-1 +] row-cooperative 4-3; row-cooperative scan 2-1; two-row fold
    # 2-1 AND 4-3): each forced for every size
    for coop_max, coop2d_max, coop4_max, coop43_max, coop2d43_max in (
            ("0", "0", "0", "0", "0"), ("1000000000", "0", "0", "1000000000", "0"), ("0", "0", "1000000000", "1", "0"),
            ("0", "1000000000", "0", "0", "1000000000")):
        env = dict(os.environ, ANEMOI_COOP_MAX=coop_max, ANEMOI_COOP2D_MAX=coop2d_max, ANEMOI_COOP4_MAX=coop4_max,
                   ANEMOI_COOP43_MAX=coop43_max, ANEMOI_COOP2D43_MAX=coop2d43_max)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "ok" in out.stdout, (coop_max, coop2d_max, coop4_max, coop43_max, coop2d43_max,
                                                            out.stdout[-1500:], out.stderr[-1500:])


def test_concurrent_callers(A, oracle, params):
    """The C-ABI is re-entrant: four host threads hammer different entry points (different fields,
    so the lazily created per-device constant tables race too) and every result must be exact."""
    import threading
    fid_b, fid_j = FIELD_IDS.index("pallas"), FIELD_IDS.index("ed_on_bls12_377")
    p_b, p_j = int(params["pallas"]["modulus"]), int(params["ed_on_bls12_377"]["modulus"])
    st_b = rand_elems(oracle, fid_b, p_b, 2 * 3000, 1).reshape(3000, 2, 4)
    st_j = rand_elems(oracle, fid_j, p_j, 4 * 900, 2).reshape(900, 4, 4)
    msgs = np.random.default_rng(3).integers(0, 256, size=(200, 95), dtype=np.uint8)
    exp_b = oracle.compress_batch(fid_b, 2, st_b, threads=8)
    exp_j = oracle.compress_batch(fid_j, 4, st_j, threads=8)
    exp_h = oracle.hash_bytes_batch(fid_j, 2, msgs, threads=8)
    errors = []

    def worker(kind):
        try:
            for _ in range(3):
                if kind == 0:
                    assert (A.Anemoi("pallas", 2).compress_batch(st_b) == exp_b).all()
                elif kind == 1:
                    assert (A.Anemoi("ed_on_bls12_377", 4).compress_batch(st_j) == exp_j).all()
                elif kind == 2:
                    assert (A.Anemoi("ed_on_bls12_377", 2).hash_batch(msgs) == exp_h).all()
                else:
                    assert (A.Anemoi("pallas", 2).merkle_root(st_b[:256, 0], 8)
                            == oracle.merkle_root(fid_b, st_b[:256, 0], 8)).all()
        except Exception as e:  # noqa: BLE001
            errors.append((kind, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the driver's keys (+ roofline, alu); small batch, no CPU leg"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--batch-log2", "12", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["value"] > 0 and abs(d["value"] - 4096 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-6
