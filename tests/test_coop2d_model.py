"""The two-row fold product (csrc/coop2d.h) on the CPU: (1) its executable specification tools/coop2d_model.py against
big-integer arithmetic on random and adversarial limbs, with 32-/64-bit overflow checks at every step; (2) the
GENERATED assembly (tools/gen_coop2d_asm.py -> csrc/coop2d_asm_gen.h) executed instruction by instruction on a 64-lane
interpreter with the DPP controls, row / bank masks and v_permlane16_swap modelled -- multiplication and squaring runs,
both layouts, two elements per wavefront that must not see each other; (3) the hazard distances of the emitted text
re-checked independently of the generator's own padding; (4) the committed header is what the generator writes today.
What a CPU cannot model is timing; what it can is every bit."""
import os
import random
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import coop2d_model as M  # noqa: E402
import gen_coop2d_asm as G  # noqa: E402
import gen_params as GP  # noqa: E402

M32, M64 = (1 << 32) - 1, (1 << 64) - 1


@pytest.fixture(scope="module")
def moduli(params):
    from conftest import FIELD_IDS
    return {f: int(params[f]["modulus"]) for f in FIELD_IDS}


def layouts(moduli):
    return {f: M.Layout(p, GP.fold_w(p)) for f, p in moduli.items()}


def operands(L, rng, bound_bits=35):
    """limb lists a product may meet: all limbs at their maximum (2^W + 31), values next to the bound, p - 1, 0, random"""
    bound = min(L.R - 1, (1 << bound_bits) * L.p)
    full = [L.M + 32] * L.NL
    full[-1] = min(bound >> (L.W * (L.NL - 1)), L.M + 32)
    out = [full + [0] * (16 - L.NL), L.limbs(bound), L.limbs(L.p - 1), L.limbs(0), L.limbs(1)]
    out += [L.limbs(rng.randrange(bound)) for _ in range(4)]
    return out


def test_model_against_big_integers(moduli):
    """every field's layout: a b R'^-1 mod p at the value level, output limbs < 2^W + 2^5, nothing overflows for inputs
    up to 2^35 p with every limb at its maximum"""
    for f, L in layouts(moduli).items():
        rng = random.Random(hash(f) & 0xffff)
        assert L.H >= 1 << 34 and L.NL in (11, 15)
        ops = operands(L, rng)
        for a in ops:
            for b in ops:
                r = M.mul(L, a, b)
                assert (L.value(r) * L.R - L.value(a) * L.value(b)) % L.p == 0, f
                assert all(x <= L.M + 32 for x in r[:L.NL]) and all(x == 0 for x in r[L.NL:]), f
                assert L.value(r) < (1 << 33) * L.p, f
                if L.NL <= 13:      # the four-row form (one element per wavefront) gives the same LIMBS
                    assert M.mul4(L, a, b) == r, f


def test_bounds_of_a_round(moduli, params):
    """The fold arithmetic never reduces below 2p: it relies on R'/p >= 2^39 and on ONE settle of x and y per round
    (coop_permutation, after the linear layer).  Walk the bounds of a round (in units of p, exact fractions) the way
    coop_permutation / coop_flystel compute it and check: every subtrahend is <= the pad of `sub`, every value stays
    below R', the Jive output's digit-serial product ends below 2p -- and run the model's product with every limb at its
    maximum at the largest bounds a product meets (64-bit column sums, 32-bit carries)."""
    from fractions import Fraction as Fr
    import math
    for f, L in layouts(moduli).items():
        g = params[f]["beta"]
        H = Fr(L.R, L.p)
        fold = L.NL * (L.M + 33) + 2          # sum_k t_k C_k < NL (2^W + 32) p, plus the carries
        K = Fr(1 << math.ceil(math.log2(g) + 32.5))
        cap = H                                # values must stay below R' = H p
        worst_in = [Fr(0)]

        def mul(a, b):
            worst_in[0] = max(worst_in[0], a, b)
            return a * b / H + fold

        S = mul(cap / 2, Fr(1))                # settle(v) = v * (R' mod p): whatever v was (< R'), the result is ~fold
        x = y = S
        for _ in range(3):                     # the bounds reach their fixed point at once; three rounds to be sure
            x, y = x + 1, y + 1                # ark_layer: canonical constants
            y = y + x                          # mds_layer, arm 1 (src/traits.rs:136-142)
            x = x + y
            assert x < cap and y < cap, (f, float(x), float(cap))
            x, y = mul(x, Fr(1)), mul(y, Fr(1))          # settle
            t = mul(y, y)
            gt = g * t
            assert gt <= K, (f, "first subtrahend", float(gt), float(K))
            x = x + K                                    # x - g y^2 + pad
            pw = mul(x, x)
            for _ in range(40):                          # table powers and the exponentiation: products of products
                pw = mul(max(pw, x), max(pw, x))
            e = pw
            assert e <= K, (f, "second subtrahend", float(e), float(K))
            y = y + K                                    # y - x^(1/alpha) + pad
            t = mul(y, y)
            x = x + g * t + 1
            assert x < cap and y < cap
        jive = 2 * (x + y)                               # state[0] + state[1] + elems (all four below max(x, y))
        assert jive / H + 1 < 2, (f, "mul_exact for to_abi must end below 2p")
        # Anemoi-4-3 on the same arithmetic (a column per row pair): the linear layer of two columns, mds_layer arm 2
        # (src/traits.rs:143-157), with g * v taken as a PRODUCT (mul_g_settled) so that the sums stay small
        x0 = x1 = y0 = y1 = S
        for _ in range(3):
            x0, x1, y0, y1 = x0 + 1, x1 + 1, y0 + 1, y1 + 1
            x0 = x0 + mul(x1, Fr(1))                     # s0 += g s1
            x1 = x1 + mul(x0, Fr(1))                     # s1 += g s0
            y1 = y1 + mul(y0, Fr(1))                     # s3 += g s2
            y0 = y0 + mul(y1, Fr(1))                     # s2 += g s3
            y0, y1 = y1, y0                              # swap(s2, s3)
            y0, y1 = y0 + x0, y1 + x1                    # s2 += s0 ; s3 += s1
            x0, x1 = x0 + y0, x1 + y1                    # s0 += s2 ; s1 += s3
            assert max(x0, x1, y0, y1) < cap, (f, "4-3 linear layer", float(max(x0, x1, y0, y1)), float(cap))
            cols = []
            for xc, yc in ((x0, y0), (x1, y1)):          # settle, then the S-box of each column as above
                xc, yc = mul(xc, Fr(1)), mul(yc, Fr(1))
                t = mul(yc, yc)
                assert g * t <= K
                xc = xc + K
                pw = mul(xc, xc)
                for _ in range(40):
                    pw = mul(max(pw, xc), max(pw, xc))
                assert pw <= K
                yc = yc + K
                t = mul(yc, yc)
                xc = xc + g * t + 1
                assert xc < cap and yc < cap
                cols.append((xc, yc))
            (x0, y0), (x1, y1) = cols
        jive4 = 2 * (x0 + y0 + x1 + y1)                  # compress_k(., 4): all four sums and the elements
        assert jive4 / H + 1 < 2, (f, "mul_exact for to_abi must end below 2p (4-3)")
        bits = math.ceil(math.log2(worst_in[0]))
        assert (1 << bits) * L.p < L.R
        rng = random.Random(5)
        ops = operands(L, rng, bound_bits=bits)
        for a in ops[:3]:
            for b in ops[:3]:
                r = M.mul(L, a, b)                       # raises Overflow if a column sum or a carry does not fit
                assert (L.value(r) * L.R - L.value(a) * L.value(b)) % L.p == 0


def test_model_notices_an_overflow(moduli):
    """sanity of the checker: limbs twice as large as the layout allows must trip it"""
    L = layouts(moduli)["bls12_381"]
    big = [4 * L.M] * L.NL + [0]
    with pytest.raises(M.Overflow):
        M.mul(L, big, big)


# ---- the generated assembly on a 64-lane interpreter ---------------------------------------------------------------------

def parse_dpp(text):
    ctrl = {"row_mask": 0xf, "bank_mask": 0xf, "bound": False, "kind": None, "n": 0}
    for k in ("row_shr", "row_shl", "row_ror", "row_newbcast"):
        m = re.search(k + r":(\d+)", text)
        if m:
            ctrl["kind"], ctrl["n"] = k, int(m.group(1))
    for k in ("row_mask", "bank_mask"):
        m = re.search(k + r":0x([0-9a-f]+)", text)
        if m:
            ctrl[k] = int(m.group(1), 16)
    ctrl["bound"] = "bound_ctrl:1" in text
    return ctrl


def dpp_apply(old, src, ctrl, fn):
    """new destination: fn(source lane value) in enabled lanes, `old` elsewhere"""
    out = list(old)
    for lane in range(64):
        row, l = lane >> 4, lane & 15
        if not (ctrl["row_mask"] >> row) & 1 or not (ctrl["bank_mask"] >> (l >> 2)) & 1:
            continue
        k, n = ctrl["kind"], ctrl["n"]
        if k == "row_shr":
            sl = l - n
        elif k == "row_shl":
            sl = l + n
        elif k == "row_ror":
            sl = (l - n) % 16
        else:
            sl = n
        if 0 <= sl < 16:
            out[lane] = fn(src[(row << 4) + sl], lane)
        elif ctrl["bound"]:
            out[lane] = fn(0, lane)
    return out


def run_asm(lines, opnd):
    """opnd: {"%0": [64 values], ..., "%k": int for an SGPR}.  Returns the final %0."""
    v, sreg = {}, {}
    scc = 0

    def get(tok):
        tok = tok.strip()
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            lo, hi = v["v%s" % m.group(1)], v["v%d" % (int(m.group(1)) + 1)]
            return [a | (b << 32) for a, b in zip(lo, hi)]
        if tok in opnd:
            x = opnd[tok]
            return list(x) if isinstance(x, list) else [x] * 64
        if tok.startswith("v"):
            return list(v[tok])
        return [int(tok, 0)] * 64

    def put(tok, vals):
        tok = tok.strip()
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            v["v%s" % m.group(1)] = [x & M32 for x in vals]
            v["v%d" % (int(m.group(1)) + 1)] = [(x >> 32) & M32 for x in vals]
        elif tok in opnd:
            opnd[tok] = [x & M32 for x in vals]
        else:
            v[tok] = [x & M32 for x in vals]

    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    pc, steps = 0, 0
    while pc < len(lines):
        ln = lines[pc]
        pc += 1
        steps += 1
        assert steps < 100000
        if ln.endswith(":") or ln.startswith("s_nop") or ln.startswith(".p2align") or ln.startswith("v_nop"):
            continue               # (v_nop_e64: the 8-byte stand-in for a lone s_nop 0, tools/asm_grid.py)
        op, _, rest = ln.partition(" ")
        op = re.sub(r"_e(32|64)$", "", op)          # the encoding (4 / 8 bytes) is layout, not semantics
        ctrl = None
        m = re.search(r"\s(row_(?:shr|shl|ror|newbcast):.*)$", rest)
        if m:
            ctrl, rest = parse_dpp(m.group(1)), rest[:m.start()]
        args = [x.strip() for x in re.split(r",\s*(?![^\[]*\])", rest)] if rest.strip() else []
        if op == "s_sub_u32":
            opnd[args[0]] = (opnd[args[1]] - int(args[2], 0)) & M32
        elif op == "s_cmp_lg_u32":
            scc = int(opnd[args[0]] != int(args[1], 0))
        elif op == "s_cmp_eq_u32":
            scc = int(opnd[args[0]] == int(args[1], 0))
        elif op == "s_cbranch_scc1":
            if scc:
                pc = labels[args[0][:-1]]       # "1b" / "2f" -> label "1" / "2"
        elif op == "s_branch":
            pc = labels[args[0][:-1]]
        elif op == "v_mov_b32":
            put(args[0], get(args[1]))
        elif op == "v_mov_b32_dpp":
            put(args[0], dpp_apply(get(args[0]) if args[0] in v or args[0] in opnd else [0] * 64, get(args[1]), ctrl, lambda s, lane: s))
        elif op == "v_add_u32_dpp":
            src1 = get(args[2])
            old = get(args[0]) if args[0] in v or args[0] in opnd else [0] * 64
            new = dpp_apply(old, get(args[1]), ctrl, lambda s, lane: s + src1[lane])
            if max(new) > M32:
                raise M.Overflow(ln)
            put(args[0], new)
        elif op == "v_add_u32":
            s = [x + y for x, y in zip(get(args[1]), get(args[2]))]
            if max(s) > M32:
                raise M.Overflow(ln)
            put(args[0], s)
        elif op == "v_and_b32":
            put(args[0], [x & y for x, y in zip(get(args[1]), get(args[2]))])
        elif op == "v_lshrrev_b32":
            put(args[0], [y >> (x & 31) for x, y in zip(get(args[1]), get(args[2]))])
        elif op == "v_alignbit_b32":
            hi, lo, sh = get(args[1]), get(args[2]), int(args[3], 0)
            full = [((h << 32) | l) >> sh for h, l in zip(hi, lo)]
            if max(full) > M32:
                raise M.Overflow(ln)       # the kernel relies on the carry fitting 32 bits
            put(args[0], full)
        elif op == "v_mad_u64_u32":
            s0, s1, s2 = get(args[2]), get(args[3]), get(args[4])
            val = [x * y + z for x, y, z in zip(s0, s1, s2)]
            if max(val) > M64:
                raise M.Overflow(ln)
            put(args[0], val)
        elif op == "v_lshl_add_u64":
            assert int(args[2], 0) == 0
            val = [x + y for x, y in zip(get(args[1]), get(args[3]))]
            if max(val) > M64:
                raise M.Overflow(ln)
            put(args[0], val)
        elif op == "v_permlane32_swap_b32":
            a, b = get(args[0]), get(args[1])
            put(args[0], a[:32] + b[:32])          # rows 2, 3 of vdst <-> rows 0, 1 of src0
            put(args[1], a[32:] + b[32:])
        elif op == "v_permlane16_swap_b32":
            a, b = get(args[0]), get(args[1])
            na, nb = list(a), list(b)
            for pair in (0, 2):           # odd rows of vdst <-> even rows of src0
                for l in range(16):
                    na[((pair + 1) << 4) + l], nb[(pair << 4) + l] = b[(pair << 4) + l], a[((pair + 1) << 4) + l]
            put(args[0], na)
            put(args[1], nb)
        else:
            raise AssertionError("instruction not modelled: " + ln)
    return opnd["%0"]


def wave(rows):
    """two elements per wavefront: element e on rows 2e and 2e+1 (both rows hold the same limbs)"""
    return [rows[lane >> 5][lane & 15] for lane in range(64)]


@pytest.mark.parametrize("field", ["jubjub", "bn_254", "ed_on_bls12_377", "vesta", "bls12_381", "bls12_377"])
def test_generated_assembly_on_the_lane_interpreter(moduli, field):
    L = layouts(moduli)[field]
    rng = random.Random(7)
    names = {k: G.operand_names(L.NL, k) for k in G.KINDS}
    lines = {k: G.gen_product(L.NL, L.W, k)[0] for k in G.KINDS}
    ct = [[(L.C[2 * q + ((lane >> 4) & 1)] >> (L.W * (lane & 15))) & L.M if (2 * q + ((lane >> 4) & 1) < L.NL and (lane & 15) < L.NL) else 0
           for lane in range(64)] for q in range(L.Q)]
    extra = {}
    if L.NL > 13:
        extra = {"MTOP": [0xffffffff if (lane & 15) == 15 else L.M for lane in range(64)],
                 "ONLY15": [0xffffffff if (lane & 15) == 15 else 0 for lane in range(64)]}

    def run(kind, a2, b2=None, n=None):
        nm = names[kind]
        opnd = {nm["A"]: wave(a2)}
        if b2 is not None:
            opnd[nm["B"]] = wave(b2)
        if n is not None:
            opnd[nm["CNT"]] = n
        opnd.update({nm["CT"][q]: ct[q] for q in range(L.Q)})
        opnd.update({nm[k]: val for k, val in extra.items()})
        return run_asm(lines[kind], opnd)

    def both_rows(r, e, want, what):
        for row in (2 * e, 2 * e + 1):
            assert r[row * 16:(row + 1) * 16] == want, (field, what, e, row)

    ops = operands(L, rng)
    for trial in range(12):
        a2 = [ops[(trial * 3 + e) % len(ops)] for e in range(2)]          # the two elements of the wavefront differ
        b2 = [ops[(trial * 5 + 2 * e + 1) % len(ops)] for e in range(2)]
        r = run("mul", a2, b2)
        for e in range(2):
            both_rows(r, e, M.mul(L, a2[e], b2[e]), ("mul", trial))      # the specification, limb for limb
        for n in (1, 2, 4):      # a run of n squarings = n products of the model; then the multiplication
            r = run("sqr_run", a2, n=n)
            r2 = run("sqr_mul", a2, b2, n=n)
            for e in range(2):
                want = a2[e]
                for _ in range(n):
                    want = M.mul(L, want, want)
                both_rows(r, e, want, ("sqr_run", trial, n))
                both_rows(r2, e, M.mul(L, want, b2[e]), ("sqr_mul", trial, n))


@pytest.mark.parametrize("field", ["jubjub", "vesta", "bls12_381"])
def test_same_value_operands_and_what_register_aliasing_would_do(moduli, field):
    """`a` and `b` holding the SAME VALUE -- the leading-run doubling of Pallas / Vesta does `tmp = acc; acc =
    sqr_mul(acc, n, tmp)` -- is fine in separate registers and WRONG if the register allocator merged them, because the
    squaring statements rewrite %0 long before they read their last input.  The interpreter cannot see register
    allocation, so: (i) same value, separate registers = the model; (ii) the same statement with `b` ALIASED to `a`'s
    register gives something else (that is what a plain "+v"(a) permitted); (iii) the generator derives the need for an
    early-clobber `a` from its own text and the committed header declares it."""
    L = layouts(moduli)[field]
    rng = random.Random(3)
    nm = G.operand_names(L.NL, "sqr_mul")
    lines = G.gen_product(L.NL, L.W, "sqr_mul")[0]
    ct = [[(L.C[2 * q + ((lane >> 4) & 1)] >> (L.W * (lane & 15))) & L.M if (2 * q + ((lane >> 4) & 1) < L.NL and (lane & 15) < L.NL) else 0
           for lane in range(64)] for q in range(L.Q)]
    base = {nm["CT"][q]: ct[q] for q in range(L.Q)}
    if L.NL > 13:
        base[nm["MTOP"]] = [0xffffffff if (lane & 15) == 15 else L.M for lane in range(64)]
        base[nm["ONLY15"]] = [0xffffffff if (lane & 15) == 15 else 0 for lane in range(64)]
    a2 = operands(L, rng)[5:7]
    for n in (1, 3):
        want = []
        for e in range(2):
            w = a2[e]
            for _ in range(n):
                w = M.mul(L, w, w)
            want.append(M.mul(L, w, a2[e]))
        opnd = dict(base)
        opnd.update({nm["A"]: wave(a2), nm["B"]: wave(a2), nm["CNT"]: n})
        r = run_asm(lines, opnd)
        for e in range(2):
            assert r[2 * e * 16:(2 * e + 1) * 16] == want[e], (field, n)
        aliased = [ln.replace(nm["B"], nm["A"]) for ln in lines]        # b lives in a's register
        opnd = dict(base)
        opnd.update({nm["A"]: wave(a2), nm["CNT"]: n})
        try:
            r = run_asm(aliased, opnd)
        except M.Overflow:
            r = None
        assert r is None or r[0:16] != want[0], (field, n, "aliasing would have gone unnoticed")


def test_early_clobber_of_the_in_out_operand():
    text = open(os.path.join(ROOT, "anemoi-rust_amd", "csrc", "coop2d_asm_gen.h")).read()
    for nl, W, rows in [(nl, W, r) for nl, W in G.LAYOUTS for r in ((2, 4) if nl <= 13 else (2,))]:
        body = text[text.index("template <> struct AsmCoop2d<%d, %d, %d> {" % (nl, W, rows)):]
        body = body[:body.index("\n};")]
        for kind in G.KINDS:
            lines = G.gen_product(nl, W, kind, rows)[0]
            late = G.inputs_read_after_first_write(lines, nl, kind, rows)
            fn = body[body.index("uint32_t %s(" % kind):]
            outs = re.search(r'\n        : ("\+&?v"\(a\)[^\n]*)\n', fn).group(1)
            assert (kind == "mul") == (not late), (nl, rows, kind)       # only the plain product writes its result last
            assert outs.startswith('"+&v"(a)' if late else '"+v"(a)'), (nl, rows, kind, outs)


@pytest.mark.parametrize("field", ["jubjub", "bn_254", "ed_on_bls12_377", "pallas"])
def test_generated_four_row_assembly_on_the_lane_interpreter(moduli, field):
    """One element on all four rows of the wavefront (11-limb fields): multiplication, squaring runs, fused steps."""
    L = layouts(moduli)[field]
    rng = random.Random(11)
    Q4 = (L.NL + 3) // 4
    names = {k: G.operand_names(L.NL, k, rows=4) for k in G.KINDS}
    lines = {k: G.gen_product(L.NL, L.W, k, rows=4)[0] for k in G.KINDS}
    ct = [[(L.C[4 * q + ((lane >> 4) & 3)] >> (L.W * (lane & 15))) & L.M if (4 * q + ((lane >> 4) & 3) < L.NL and (lane & 15) < L.NL) else 0
           for lane in range(64)] for q in range(Q4)]

    def run(kind, a, b=None, n=None):
        nm = names[kind]
        opnd = {nm["A"]: [a[lane & 15] for lane in range(64)]}
        if b is not None:
            opnd[nm["B"]] = [b[lane & 15] for lane in range(64)]
        if n is not None:
            opnd[nm["CNT"]] = n
        opnd.update({nm["CT"][q]: ct[q] for q in range(Q4)})
        r = run_asm(lines[kind], opnd)
        assert r[0:16] == r[16:32] == r[32:48] == r[48:64], (field, kind)   # every row ends with the whole result
        return r[0:16]

    ops = operands(L, rng)
    for trial in range(10):
        a, b = ops[(trial * 3) % len(ops)], ops[(trial * 5 + 1) % len(ops)]
        assert run("mul", a, b) == M.mul(L, a, b), (field, trial)
        for n in (1, 3) + ((G.UNROLL, G.UNROLL + 2) if trial < 2 else ()):   # (the longer runs wrap round the unrolled copies)
            want = a
            for _ in range(n):
                want = M.mul(L, want, want)
            assert run("sqr_run", a, n=n) == want, (field, trial, n)
            assert run("sqr_mul", a, b, n=n) == M.mul(L, want, b), (field, trial, n)


def test_hazard_distances_of_the_emitted_text():
    """Independent of the generator's padding: a VGPR written by a VALU instruction is not read through DPP (source or
    destination of a *_dpp instruction) or by v_permlane16_swap within the next two issue slots -- also across the
    back-edge of the squaring loop (checked by unrolling the body twice)."""
    for nl, W, rows in [(nl, W, r) for nl, W in G.LAYOUTS for r in ((2, 4) if nl <= 13 else (2,))]:
        for kind in G.KINDS:
            lines = G.gen_product(nl, W, kind, rows)[0]
            # A run of squarings is already laid out as UNROLL >= 2 copies of the body, each followed by an exit branch:
            # read as straight-line code that checks body -> body (which is also the back-edge: `s_branch 1b` leads from
            # the last copy to the first) and, without the s_branch, last body -> exit branch -> what follows the run
            # (every copy ends the same way, so that is every exit).  A forward branch is checked as fall-through: its
            # target sees at least the registers written before the branch.
            assert G.UNROLL >= 2
            lines = [l for l in lines if not l.startswith(".p2align") and not l.startswith("s_branch")]
            lines = [l for l in lines if not l.endswith(":")]
            slot, wrote = 0, {}
            for ln in lines:
                op, _, rest = ln.partition(" ")
                if op == "s_nop":
                    slot += int(rest) + 1
                    continue
                if ln.endswith(":"):
                    continue
                if op.startswith("v_nop"):
                    slot += 1
                    continue
                op = re.sub(r"_e(32|64)$", "", op)
                regs = re.findall(r"v\[\d+:\d+\]|v\d+|%\d+", rest.split(" row_")[0])
                flat = []
                for r in regs:
                    m = re.fullmatch(r"v\[(\d+):(\d+)\]", r)
                    flat.append(["v%d" % i for i in range(int(m.group(1)), int(m.group(2)) + 1)] if m else [r])
                sensitive = []
                if op.endswith("_dpp"):
                    sensitive = flat[0] + flat[1]            # destination (old value) and the DPP source
                elif op.startswith("v_permlane"):
                    sensitive = flat[0] + flat[1]
                for r in sensitive:
                    if r in wrote:
                        assert slot - wrote[r] - 1 >= 2, (nl, kind, ln, r)
                if op.startswith("v_"):
                    written = flat[0] + (flat[1] if op.startswith("v_permlane") else [])
                    for r in written:
                        wrote[r] = slot
                slot += 1


def test_statements_sit_on_the_8_byte_fetch_grid():
    """A lone wavefront pays ~1 cycle for every 8-byte instruction that straddles an 8-byte boundary
    (tools/ubench/lone_wave_fetch.hip), so the generator lays the statements out on an 8-byte grid.  Checked with the
    assembler itself: every line's encoded size is what the generator assumed, and every 8-byte instruction starts at a
    multiple of 8 from the statement's (aligned) start."""
    import shutil
    import subprocess
    mc = shutil.which("llvm-mc") or "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        pytest.skip("no llvm-mc")
    for nl, W, rows in [(nl, W, r) for nl, W in G.LAYOUTS for r in ((2, 4) if nl <= 13 else (2,))]:
        for kind in G.KINDS:
            lines, _, info = G.gen_product(nl, W, kind, rows)
            assert info["off_grid"] == 0, (nl, kind)
            assert lines[0] == ".p2align 3"
            cnt = G.operand_names(nl, kind, rows).get("CNT")
            text = []
            for ln in lines:
                if cnt:
                    ln = re.sub(re.escape(cnt) + r"(?!\d)", "s40", ln)
                text.append(re.sub(r"%(\d+)", lambda m: "v%d" % (200 + int(m.group(1))), ln))
            out = subprocess.run([mc, "-triple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-show-encoding"],
                                 input="\n".join(text) + "\n", capture_output=True, text=True, check=True).stdout
            sizes = [len(m.group(1).split(",")) for m in re.finditer(r"encoding: \[([^\]]*)\]", out)]
            insts = [ln for ln in lines if not ln.endswith(":") and not ln.startswith(".")]
            assert len(sizes) == len(insts), (nl, kind, len(sizes), len(insts))
            off = 0
            for ln, size in zip(insts, sizes):
                assert size == G.enc_size(ln), (nl, kind, ln, size)
                assert size == 4 or off % 8 == 0, (nl, kind, ln, off)
                off += size


def test_committed_header_is_what_the_generator_writes():
    import io
    import contextlib
    path = os.path.join(ROOT, "anemoi-rust_amd", "csrc", "coop2d_asm_gen.h")
    before = open(path).read()
    with contextlib.redirect_stdout(io.StringIO()):
        G.main()
    assert open(path).read() == before, "csrc/coop2d_asm_gen.h is stale: run tools/gen_coop2d_asm.py"


def test_issue_slots_of_the_generated_statements_do_not_regress():
    """The latency of a compression is the issue-slot count of its products: pin what the generator
    achieves today -- 78 / 79 slots per multiplication / squaring on 11 limbs, 96 / 97 on 15, six wait slots left per
    squaring -- so that an edit of the generator that costs a slot is noticed here, not on the GPU."""
    want = {(11, 2): (78, 79), (15, 2): (96, 97), (11, 4): (83, 86)}
    for (nl, rows), (mul_slots, sqr_slots) in want.items():
        W = dict(G.LAYOUTS)[nl]
        assert G.gen_product(nl, W, "mul", rows)[2]["slots"] <= mul_slots, (nl, rows)
        for kind in ("sqr_run", "sqr_mul"):
            lines, _, info = G.gen_product(nl, W, kind, rows)
            assert info["slots"] <= sqr_slots, (nl, rows, kind, info["slots"])
            assert info["off_grid"] == 0
        if rows == 2:
            assert G.gen_product(nl, W, "sqr_run", rows)[2]["nops"] <= 6 * G.UNROLL


def test_the_fetch_grid_pass_is_optimal_on_small_texts():
    """tools/asm_grid.align8 against brute force: random short instruction lists (4-byte promotable / splittable /
    fixed, 8-byte) -- the dynamic programme leaves as few 8-byte instructions off the grid as any choice of encodings."""
    import itertools
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import asm_grid
    pool = ["v_add_u32 v1, v2, v3", "s_sub_u32 s4, s4, 1", "s_nop 1", "s_nop 0", "v_mad_u64_u32 v[2:3], vcc, v4, v5, v[2:3]",
            "v_mov_b32_dpp v1, v2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", "v_and_b32 v1, 0x7ffffff, v2",
            "s_cbranch_scc1 1b", "v_lshrrev_b32 v1, 27, v2"]
    rng = random.Random(12)
    for trial in range(200):
        text = [rng.choice(pool) for _ in range(rng.randrange(1, 9))]
        for vnop in (False, True):
            got_lines, got = asm_grid.align8(text, vnop=vnop)
            options = [asm_grid._variants(ln, vnop) for ln in text]
            best = None
            for pick in itertools.product(*[range(len(o)) for o in options]):
                off, cost = 0, 0
                for o, i in zip(options, pick):
                    size = o[i][1]
                    cost += 1 if size == 8 and off % 8 else 0
                    off += size
                best = cost if best is None else min(best, cost)
            assert got == best, (text, vnop, got, best)
            off = bad = 0
            for ln in got_lines:
                size = asm_grid.enc_size(ln)
                bad += 1 if size == 8 and off % 8 else 0
                off += size
            assert bad == got, (text, got_lines)
