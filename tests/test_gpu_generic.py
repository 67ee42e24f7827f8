"""GPU parity for instances given by run-time trait constants: the reference's
`mds_layer` arms for 3..6 columns, its matrix arm, `exp_by_alpha` -- code the reference carries but no
shipped instance reaches, so it has NO reference KATs.  Pinning:
  * the generic kernels with the SHIPPED constants (1 and 2 columns) must reproduce the reference's own
    KATs, with the hard-coded arm and with the arm's matrix given explicitly;
  * wider instances are compared with the statement-by-statement restatement in oracle/anemoi_ref.py,
    which tests/test_oracle.py ties to the matrix form.
Bit-exact.  Run on the GPU box: pytest -m gpu.
"""
import random

import numpy as np
import pytest

from conftest import FIELD_IDS, INSTANCES, inst_key

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


def ints(v):
    return [int(x) for x in v]


def make_instance(A, R, oracle, field, cols, rounds, seed, with_matrix=None):
    """random round constants (and matrix): returns (GPU instance, Python oracle instance, enc, dec)"""
    fid = FIELD_IDS.index(field)
    base = R.Instance(field, 2)
    rng = random.Random(seed)
    C = [rng.randrange(base.p) for _ in range(cols * rounds)]
    D = [rng.randrange(base.p) for _ in range(cols * rounds)]
    M = None
    if with_matrix == "random":
        M = [rng.randrange(base.p) for _ in range(cols * cols)]
    elif with_matrix == "builtin":
        M = R.builtin_mds_matrix(cols, base.g, base.p)
    enc = lambda v: oracle.ints_to_mont(fid, [int(x) for x in v])
    dec = lambda a: oracle.mont_to_ints(fid, np.asarray(a, dtype=np.uint64).reshape(-1, base.limbs))
    gpu = A.GenericAnemoi(field, cols, rounds, enc(C), enc(D), None if M is None else enc(M))
    ref = R.GenericInstance(field, cols, rounds, C, D, M)
    return gpu, ref, enc, dec


@pytest.mark.parametrize("field,width", INSTANCES)
@pytest.mark.parametrize("matrix", [None, "builtin"])
def test_shipped_instances_through_the_generic_path(A, R, oracle, kats, field, width, matrix):
    k, I = kats[inst_key(field, width)], R.Instance(field, width)
    fid, c = FIELD_IDS.index(field), width // 2
    enc = lambda v: oracle.ints_to_mont(fid, [int(x) for x in v])
    dec = lambda a: oracle.mont_to_ints(fid, np.asarray(a, dtype=np.uint64).reshape(-1, I.limbs))
    M = None if matrix is None else enc(R.builtin_mds_matrix(c, I.g, I.p))
    g = A.GenericAnemoi(field, c, I.rounds, enc(I.C), enc(I.D), M)
    if matrix == "builtin":  # the library's own idea of the arm's matrix
        assert (A.builtin_mds_matrix(field, c) == M).all()
    # test_anemoi_jive (hasher.rs:231) and compress_k(., 4)
    st = np.stack([enc(ints(a)) for a in k["jive"]["in"]])
    got = g.compress_k_batch(st, 2)
    for row, b in zip(got, k["jive"]["out"]):
        assert dec(row) == ints(b)
    if width == 4:
        st = np.stack([enc(ints(a)) for a in k["jive_k4"]["in"]])
        for row, b in zip(g.compress_k_batch(st, 4), k["jive_k4"]["out"]):
            assert dec(row) == ints(b)
    # test_anemoi_hash (hasher.rs:123): inputs of different lengths
    for a, b in zip(k["hash_field"]["in"], k["hash_field"]["out"]):
        e = enc(ints(a)).reshape(1, -1, I.limbs)
        assert dec(g.hash_field_batch(e, I.rate)) == [int(b)]
    # test_anemoi_hash_bytes (hasher.rs:202)
    for a, b in zip(k["hash_bytes"]["in_hex"], k["hash_bytes"]["out"]):
        m = np.frombuffer(bytes.fromhex(a), dtype=np.uint8).reshape(1, -1)
        assert dec(g.hash_batch(m, I.rate)) == [int(b)]


@pytest.mark.parametrize("field", FIELD_IDS)
@pytest.mark.parametrize("cols", [3, 4, 5, 6])
def test_hardcoded_arms_3_to_6_vs_oracle(A, R, oracle, field, cols):
    gpu, ref, enc, dec = make_instance(A, R, oracle, field, cols, 3, seed=100 * cols + FIELD_IDS.index(field))
    rng = random.Random(cols)
    L, w = ref.limbs, 2 * cols
    groups = 64 // cols
    for n in (1, groups - 1, groups, groups + 1, 2 * groups + 3):
        st = [[rng.randrange(ref.p) for _ in range(w)] for _ in range(n)]
        st[0] = [0] * w
        if n > 1:
            st[1] = [ref.p - 1] * w
        enc_st = np.stack([enc(s) for s in st])
        got = gpu.permutation_batch(enc_st)
        for i in range(n):
            assert dec(got[i]) == ref.permutation(list(st[i])), (field, cols, n, i)
    # Jive for every k the reference's asserts allow; the others are refused
    st = [[rng.randrange(ref.p) for _ in range(w)] for _ in range(7)]
    enc_st = np.stack([enc(s) for s in st])
    for k in range(1, w + 2):
        if k <= w and w % k == 0 and k % 2 == 0:
            got = gpu.compress_k_batch(enc_st, k)
            for i in range(7):
                assert dec(got[i]) == ref.compress_k(st[i], k)
        else:
            with pytest.raises(A.AnemoiError):
                gpu.compress_k_batch(enc_st, k)


@pytest.mark.parametrize("field", ["bls12_381", "ed_on_bls12_377", "vesta"])
@pytest.mark.parametrize("cols", [1, 2, 3, 7, 8, 16])
def test_matrix_arm_vs_oracle(A, R, oracle, field, cols):
    gpu, ref, enc, dec = make_instance(A, R, oracle, field, cols, 2, seed=7 * cols, with_matrix="random")
    rng = random.Random(cols + 1)
    w = 2 * cols
    for n in (1, 64 // cols + 2):
        st = [[rng.randrange(ref.p) for _ in range(w)] for _ in range(n)]
        got = gpu.permutation_batch(np.stack([enc(s) for s in st]))
        for i in range(n):
            assert dec(got[i]) == ref.permutation(list(st[i]))
    # the built-in matrix given explicitly == the hard-coded arm
    if cols <= 6:
        g1, r1, _, _ = make_instance(A, R, oracle, field, cols, 2, seed=99)
        g2, _, _, _ = make_instance(A, R, oracle, field, cols, 2, seed=99, with_matrix="builtin")
        st = np.stack([enc([rng.randrange(ref.p) for _ in range(w)]) for _ in range(9)])
        assert (g1.permutation_batch(st) == g2.permutation_batch(st)).all()


@pytest.mark.parametrize("field", ["bls12_377", "ed_on_bls12_377", "jubjub", "pallas"])
@pytest.mark.parametrize("cols,matrix", [(1, None), (2, "builtin"), (3, None), (5, "random")])
def test_unreduced_states_and_constants_are_taken_mod_p(A, R, oracle, field, cols, matrix):
    """The input contract of round 6 (header: UNREDUCED INPUTS) on the run-time-instance entry points: states, messages AND the
    instance's constants (ARK_C, ARK_D, the matrix) given as X + k p, up to the top of the 64 L-bit range, must give what the
    reduced values give -- permutation, Jive, hash_field; host forms and a prepared instance on device pointers."""
    import torch
    fid = FIELD_IDS.index(field)
    base = R.Instance(field, 2)
    p, L = base.p, base.limbs
    room = ((1 << (64 * L)) - 1) // p                      # X + k p fits for k < room (2 on jubjub ... 152 on bls12_377)
    rng = random.Random(17 * cols + fid)
    lift = lambda a: A.ints_to_limbs([v + rng.randrange(1, room) * p for v in A.limbs_to_ints(a)], L).reshape(np.shape(a))
    rounds = 3
    C = [rng.randrange(p) for _ in range(cols * rounds)]
    D = [rng.randrange(p) for _ in range(cols * rounds)]
    M = None if matrix is None else ([rng.randrange(p) for _ in range(cols * cols)] if matrix == "random" else R.builtin_mds_matrix(cols, base.g, p))
    enc = lambda v: oracle.ints_to_mont(fid, [int(x) for x in v])
    g_red = A.GenericAnemoi(field, cols, rounds, enc(C), enc(D), None if M is None else enc(M))
    g_raw = A.GenericAnemoi(field, cols, rounds, lift(enc(C)), lift(enc(D)), None if M is None else lift(enc(M)))
    w = 2 * cols
    st = np.stack([enc([rng.randrange(p) for _ in range(w)]) for _ in range(70)])
    raw = lift(st)
    assert (raw != st).any()
    want = g_red.permutation_batch(st)
    assert (g_raw.permutation_batch(raw) == want).all() and (g_red.permutation_batch(raw) == want).all()
    assert (g_raw.compress_k_batch(raw, 2) == g_red.compress_k_batch(st, 2)).all()
    rate = w - 1
    msgs = np.stack([enc([rng.randrange(p) for _ in range(2 * rate + 1)]) for _ in range(9)])
    assert (g_raw.hash_field_batch(lift(msgs), rate) == g_red.hash_field_batch(msgs, rate)).all()
    for inverse in (False, True):                          # exp_by_alpha / exp_by_inv_alpha, element-wise
        assert (A.exp_alpha_batch(field, raw[:, 0], inverse=inverse) == A.exp_alpha_batch(field, st[:, 0], inverse=inverse)).all()
    prep = g_raw.prepare(0)
    d = torch.from_numpy(raw.view(np.int64).reshape(-1).copy()).to("cuda:0")
    prep.permutation_dev(d.data_ptr(), len(raw), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (d.cpu().numpy().view(np.uint64).reshape(want.shape) == want).all()
    prep.close()


@pytest.mark.parametrize("field", ["bls12_377", "bn_254", "jubjub"])
def test_generic_sponge_vs_oracle(A, R, oracle, field):
    rngb = np.random.default_rng(5)
    for cols, rate in ((3, 5), (3, 1), (4, 7), (5, 4), (6, 11)):
        gpu, ref, enc, dec = make_instance(A, R, oracle, field, cols, 2, seed=cols * 31 + rate)
        rng = random.Random(rate)
        for ne in (0, 1, rate - 1, rate, rate + 1, 2 * rate, 2 * rate + 1):
            if ne < 0:
                continue
            n = 3
            el = [[rng.randrange(ref.p) for _ in range(ne)] for _ in range(n)]
            e = (np.stack([enc(m) for m in el]).reshape(n, ne, ref.limbs) if ne
                 else np.zeros((n, 0, ref.limbs), dtype=np.uint64))
            got = gpu.hash_field_batch(e, rate)
            for i in range(n):
                assert dec(got[i]) == [ref.hash_field(el[i], rate)], (field, cols, rate, ne)
        # bytes: chunking + the reference's padding rule, then the same sponge
        ch = ref.base.chunk
        for ln in (0, 1, ch, ch + 1, rate * ch, rate * ch + 3):
            msgs = rngb.integers(0, 256, size=(2, ln), dtype=np.uint8)
            got = gpu.hash_batch(msgs, rate)
            for i in range(2):
                assert dec(got[i]) == [ref.hash_field(ref.base.bytes_to_elems(msgs[i].tobytes()), rate)]


@pytest.mark.parametrize("field", FIELD_IDS)
def test_exp_by_alpha_and_its_inverse(A, R, oracle, field):
    """the reference's test_alpha: exp_by_inv_alpha(exp_by_alpha(x)) == x, and each against pow()"""
    fid, I = FIELD_IDS.index(field), R.Instance(field, 2)
    rng = random.Random(fid)
    xs = [0, 1, I.p - 1] + [rng.randrange(I.p) for _ in range(127)]
    e = oracle.ints_to_mont(fid, xs)
    up = A.exp_alpha_batch(field, e)
    assert oracle.mont_to_ints(fid, up) == [R.exp_by_alpha_chain(x, I.alpha, I.p) for x in xs]
    down = A.exp_alpha_batch(field, e, inverse=True)
    assert oracle.mont_to_ints(fid, down) == [pow(x, I.inv_alpha, I.p) for x in xs]
    assert (A.exp_alpha_batch(field, up, inverse=True) == e).all()
    assert (A.exp_alpha_batch(field, down) == e).all()


def test_generic_argument_errors(A, R, oracle):
    gpu, ref, enc, dec = make_instance(A, R, oracle, "pallas", 3, 2, seed=1)
    one = enc([1] * 6).reshape(1, 6, ref.limbs)
    with pytest.raises(A.AnemoiError):  # rate must leave a capacity element
        gpu.hash_field_batch(one, 6)
    with pytest.raises(A.AnemoiError):
        gpu.hash_field_batch(one, 0)
    zeros = lambda rows: np.zeros((rows, ref.limbs), dtype=np.uint64)
    for cols, rounds in ((17, 2), (3, 0), (3, 256)):
        bad = A.GenericAnemoi("pallas", cols, rounds, zeros(cols * rounds), zeros(cols * rounds))
        with pytest.raises(A.AnemoiError):
            bad.permutation_batch(zeros(2 * cols).reshape(1, 2 * cols, ref.limbs))
    with pytest.raises(A.AnemoiError):
        A.builtin_mds_matrix("pallas", 0)
    # more than 6 columns without a matrix: the reference's expect("NO MDS matrix specified ...")
    c7 = enc([1] * 14)
    with pytest.raises(A.AnemoiError):
        A.GenericAnemoi("pallas", 7, 2, c7, c7).permutation_batch(np.zeros((1, 14, ref.limbs), dtype=np.uint64))
    with pytest.raises(A.AnemoiError):
        A.builtin_mds_matrix("pallas", 7)


def test_generic_golden_vectors(A, R, oracle):
    """tests/golden/generic.json (tools/mint_generic_goldens.py): committed vectors for 3..7 columns"""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "generic.json")) as f:
        vecs = json.load(f)
    assert len(vecs) == 35
    for v in vecs:
        fid = FIELD_IDS.index(v["field"])
        L = R.Instance(v["field"], 2).limbs
        enc = lambda xs: oracle.ints_to_mont(fid, [int(x) for x in xs])
        dec = lambda a: oracle.mont_to_ints(fid, np.asarray(a, dtype=np.uint64).reshape(-1, L))
        g = A.GenericAnemoi(v["field"], v["num_columns"], v["num_rounds"], enc(v["ark_c"]), enc(v["ark_d"]),
                            None if v["mds"] is None else enc(v["mds"]))
        st = enc(v["state"])[None]
        assert dec(g.permutation_batch(st)) == ints(v["permutation"])
        assert dec(g.compress_k_batch(st, 2)) == ints(v["compress_k2"])
        msg = enc(v["message"]).reshape(1, -1, L)
        assert dec(g.hash_field_batch(msg, v["rate"])) == [int(v["hash_field"])]


@pytest.mark.parametrize("field,cols,matrix", [("bls12_381", 1, None), ("bn_254", 2, None), ("jubjub", 3, None),
                                               ("bls12_377", 5, None), ("vesta", 7, "random"), ("pallas", 16, "random")])
def test_prepared_instances_on_device_pointers(A, R, oracle, field, cols, matrix):
    """anemoi_generic_prepare + the `_dev` entry points (constants uploaded and converted once; buffers in HBM,
    asynchronous on a stream): permutation, Jive and both sponges against the Python restatement of the
    reference's arms (src/traits.rs:136-304) and against the host-pointer forms."""
    import torch
    gpu, ref, enc, dec = make_instance(A, R, oracle, field, cols, 3, seed=31 * cols + 5, with_matrix=matrix)
    prep = gpu.prepare()
    rng = random.Random(cols)
    L, w = ref.limbs, 2 * cols
    n = 64 // cols + 3
    st_i = [[rng.randrange(ref.p) for _ in range(w)] for _ in range(n)]
    st = np.stack([enc(s) for s in st_i])
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream()
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64).reshape(-1)).to(dev)
    back = lambda t, shape: t.cpu().numpy().view(np.uint64).reshape(shape)
    d_st = to_dev(st)
    torch.cuda.synchronize()
    # permutation, in place
    prep.permutation_dev(d_st.data_ptr(), n, stream.cuda_stream)
    stream.synchronize()
    got = back(d_st, (n, w, L))
    assert (got == gpu.permutation_batch(st)).all()
    for i in (0, n - 1):
        assert dec(got[i]) == ref.permutation(list(st_i[i]))
    # Jive for every admissible k; the reference's asserts for the others; overlapping buffers refused
    d_in = to_dev(st)
    for k in range(1, w + 2):
        ok = k <= w and w % k == 0 and k % 2 == 0
        d_out = torch.zeros(n * (w // k if ok else 1) * L, dtype=torch.int64, device=dev)
        if ok:
            prep.compress_k_dev(k, d_in.data_ptr(), d_out.data_ptr(), n, stream.cuda_stream)
            stream.synchronize()
            o = back(d_out, (n, w // k, L))
            for i in (0, n // 2, n - 1):
                assert dec(o[i]) == ref.compress_k(st_i[i], k)
        else:
            with pytest.raises(A.AnemoiError):
                prep.compress_k_dev(k, d_in.data_ptr(), d_out.data_ptr(), n, stream.cuda_stream)
    with pytest.raises(A.AnemoiError):
        prep.compress_k_dev(2, d_in.data_ptr(), d_in.data_ptr() + 8 * L, n, stream.cuda_stream)
    # sponges with rate = width - 1: elements and bytes
    rate, ne = w - 1, 2 * (w - 1) + 1
    el_i = [[rng.randrange(ref.p) for _ in range(ne)] for _ in range(n)]
    el = np.stack([enc(e) for e in el_i])
    d_el, d_dig = to_dev(el), torch.zeros(n * L, dtype=torch.int64, device=dev)
    prep.hash_field_dev(rate, d_el.data_ptr(), ne, n, d_dig.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    assert (back(d_dig, (n, L)) == gpu.hash_field_batch(el.reshape(n, ne, L), rate)).all()
    assert dec(back(d_dig, (n, L))[0]) == [ref.hash_field(el_i[0], rate)]
    msgs = np.random.default_rng(cols).integers(0, 256, size=(n, 77), dtype=np.uint8)
    d_m = torch.from_numpy(msgs).to(dev)
    prep.hash_bytes_dev(rate, d_m.data_ptr(), 77, n, d_dig.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    assert (back(d_dig, (n, L)) == gpu.hash_batch(msgs, rate)).all()
    with pytest.raises(A.AnemoiError):
        prep.hash_bytes_dev(w, d_m.data_ptr(), 77, n, d_dig.data_ptr(), stream.cuda_stream)   # rate >= width
    prep.close()
    prep.close()   # idempotent
