#!/usr/bin/env python3
"""Generate the constant headers from tests/golden/params.json (pure data -> pure data).

  oracle/anemoi_params_gen.h                   plain C tables for the CPU oracle
  anemoi-rust_amd/csrc/field_consts_gen.h      C++ constexpr tables for the HIP kernels + host

Both hold, per field: modulus p, R = 2^(64N) mod p, R^2 mod p, -p^-1 mod 2^64, DELTA and the
generator in Montgomery form, INV_ALPHA as a plain exponent, and per instance the round
constants C, D in Montgomery form.  The HIP header additionally carries sliding-window
exponentiation schedules for x^INV_ALPHA (window 2..5): any schedule computing x^INV_ALPHA mod p is
bit-identical to the reference's hard-coded addition chain (src/<f>/sbox.rs exp_by_inv_alpha),
because the result is a canonical field value.

    python tools/gen_params.py
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELD_IDS = ["bls12_381", "bls12_377", "bn_254", "ed_on_bls12_377", "jubjub", "pallas", "vesta"]
MAXL = 6
MAXR21, MAXR43 = 21, 14
# extra window digits held in VGPRs per field (0 = plain 3-bit window); see extra_digit_schedule()
XDIGITS = {"bls12_381": 2, "bls12_377": 2, "bn_254": 2, "ed_on_bls12_377": 2, "jubjub": 2}
if __import__("os").environ.get("ANEMOI_XDIGITS") is not None:
    XDIGITS = {k: int(__import__("os").environ["ANEMOI_XDIGITS"]) for k in XDIGITS}


def limbs64(v, n):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def limbs32(v, n):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def c64(vals):
    return "{" + ",".join("0x%016xULL" % v for v in vals) + "}"


def c32(vals):
    return "{" + ",".join("0x%08xu" % v for v in vals) + "}"


def sliding_window(e, k):
    """Left-to-right sliding window over odd powers x^1, x^3, .., x^(2^k-1), with one extra trick for
    exponents whose leading bits repeat (Pallas / Vesta: 1/5 mod p-1 starts with 32 nibbles 0x3).

    Returns (first_idx, steps) with steps = [(num_squarings, op)]:
        acc = T[first_idx]; for (s, op) in steps:
            op == 253:  tmp = acc                      (s = 0)
            otherwise:  acc = acc^(2^s); then  op < 253: acc *= T[op];  op == 254: acc *= tmp;  op == 255: nothing
    T[j] = x^(2j+1).  A leading run of r equal P-bit blocks (value v) is built by doubling -- u_1 = x^v,
    u_2j = u_j^(2^(P j)) * u_j -- in log2(r) multiplications instead of r - 1.
    """
    bits = bin(e)[2:]
    n = len(bits)

    def take(i):
        # longest window of <= k bits starting at i (a 1 bit) and ending in a 1 bit
        j = min(i + k, n)
        while bits[j - 1] != "1":
            j -= 1
        return j - i, int(bits[i:j], 2)

    ln, v = take(0)
    first, i, steps, pend = (v - 1) // 2, ln, [], 0
    best = (0, 0, 0)  # (saved multiplications, period, repeats as a power of two)
    for P in range(ln, 9):
        block = format(v, "0%db" % P)
        r = 1
        while bits[ln + (r - 1) * P: ln + r * P] == block:
            r += 1
        r2 = 1 << (r.bit_length() - 1)
        saved = (r2 - 1) - (r2.bit_length() - 1)
        if saved > best[0]:
            best = (saved, P, r2)
    if best[0] >= 2:
        _, P, r2 = best
        j = 1
        while j < r2:
            steps.append((0, 253))
            steps.append((P * j, 254))
            j *= 2
        i = ln + (r2 - 1) * P
    while i < n:
        if bits[i] == "0":
            pend += 1
            i += 1
        else:
            ln, v = take(i)
            steps.append((pend + ln, (v - 1) // 2))
            pend, i = 0, i + ln
    if pend:
        steps.append((pend, 255))
    return first, steps


# ---- window table with extra digits -----------------------------------------------------------------------
# The 3-bit window's table {x, x^3, x^5, x^7} is what fits in LDS next to 3-4 waves per SIMD; the spare VGPRs hold
# one or two MORE powers x^d.  Which d?  Not 9 and 11: 1/alpha mod (p - 1) of these primes has runs of repeated
# nibbles (..2233332277666662e65999999995555 for BLS12-381), and digits like 51 = 0b110011 or 17 = 0b10001
# swallow two windows at once.  For every candidate d (odd, < 2^12) an optimal left-to-right recoding over the digit
# set {1, 3, 5, 7, d1[, d2]} (dynamic programme over the bit positions, windows up to 12 bits) is priced together
# with the cheapest way to build x^d from what is at hand while the table is being built (x, x^2, x^3, x^5, x^7, the
# first extra): LOAD a ; [MUL b] ; SQR k ; [MUL r].
XBASE = (1, 3, 5, 7)
XMAXW = 12
XSRC = {1: 0, 2: 1, 3: 2, 5: 3, 7: 4}   # source ids of the build programme: x, x^2, table entries 1..3; 5 = first extra


def digit_dp(e, digits, maxw=XMAXW):
    """optimal recoding of e over the odd digit set: returns (multiplications, first digit, [(squarings, digit|None)])"""
    bits = bin(e)[2:]
    n = len(bits)
    INF = 10 ** 9
    best, choice = [INF] * (n + 1), [None] * (n + 1)
    best[n] = 0
    dset = set(digits)
    for i in range(n - 1, -1, -1):
        if bits[i] == "0":
            best[i], choice[i] = best[i + 1], (1, None)
            continue
        v = 0
        for w in range(1, maxw + 1):
            j = i + w
            if j > n:
                break
            v = (v << 1) | (bits[j - 1] == "1")
            if bits[j - 1] == "1" and v in dset and best[j] + 1 < best[i]:
                best[i], choice[i] = best[j] + 1, (w, v)
    w0, first = choice[0]
    steps, i, pend = [], w0, 0
    while i < n:
        w, v = choice[i]
        if v is None:
            pend += 1
        else:
            steps.append((pend + w, v))
            pend = 0
        i += w
    if pend:
        steps.append((pend, None))
    return best[0] - 1, first, steps


def digit_chain(d, avail):
    """cheapest programme (list of ops) that leaves x^d in the build register: ("load", a) [("mul", b)] ("sqr", k)
    [("mul", r)] with a, b, r among the exponents in `avail`"""
    best = None
    cands = []
    for a in avail:
        cands.append((a, [("load", a)]))
    for a in avail:
        for b in avail:
            if a <= b:
                cands.append((a + b, [("load", a), ("mul", b)]))
    for h, prog in cands:
        for k in range(0, XMAXW + 1):
            r = d - (h << k)
            if r < 0:
                break
            if r != 0 and r not in avail:
                continue
            ops = prog + ([("sqr", k)] if k else []) + ([("mul", r)] if r else [])
            cost = sum(o[1] if o[0] == "sqr" else (1 if o[0] == "mul" else 0) for o in ops)
            if cost and (best is None or cost < best[0]):
                best = (cost, ops)
    return best


def extra_digit_schedule(e, n_extra):
    """-> (extras [(digit, ops)], first digit, steps, products incl. the whole table build)"""
    avail0 = [1, 2, 3, 5, 7]
    base_cost, first, steps = digit_dp(e, XBASE)
    best = (base_cost + 4, [], first, steps)
    if n_extra == 0:
        return best[1], best[2], best[3], best[0]
    singles = []
    for d in range(9, 1 << XMAXW, 2):
        ch = digit_chain(d, avail0)
        if ch:
            m, f, st = digit_dp(e, XBASE + (d,))
            singles.append((m + 4 + ch[0], d, ch[1], f, st))
    singles.sort(key=lambda t: t[0])
    if singles[0][0] < best[0]:
        c, d, ops, f, st = singles[0]
        best = (c, [(d, ops)], f, st)
    if n_extra >= 2:
        for c1, d1, ops1, _, _ in singles[:10]:
            cost1 = c1 - digit_dp(e, XBASE + (d1,))[0] - 4
            for d2 in range(9, 1 << XMAXW, 2):
                if d2 == d1:
                    continue
                ch = digit_chain(d2, avail0 + [d1])
                if not ch:
                    continue
                m, f, st = digit_dp(e, XBASE + (d1, d2))
                c = m + 4 + cost1 + ch[0]
                if c < best[0]:
                    best = (c, [(d1, ops1), (d2, ch[1])], f, st)
    return best[1], best[2], best[3], best[0]


def check_extra_schedule(e, extras, first, steps, p):
    x = 0x1234567 % p
    val = {1: x, 2: x * x % p}
    for d in (3, 5, 7):
        val[d] = val[d - 2] * val[2] % p
    for d, ops in extras:
        r = None
        for op, arg in ops:
            if op == "load":
                r = val[arg]
            elif op == "sqr":
                r = pow(r, 1 << arg, p)
            else:
                r = r * val[arg] % p
        assert r == pow(x, d, p), (d, ops)
        val[d] = r
    acc = val[first]
    for s, d in steps:
        acc = pow(acc, 1 << s, p)
        if d is not None:
            acc = acc * val[d] % p
    assert acc == pow(x, e, p)


def check_schedule(e, k, first, steps, p):
    x = 0x1234567 % p
    tab = [pow(x, 2 * j + 1, p) for j in range(1 << (k - 1))]
    acc, tmp = tab[first], None
    for s, t in steps:
        if t == 253:
            assert s == 0
            tmp = acc
            continue
        acc = pow(acc, 1 << s, p)
        if t == 254:
            acc = acc * tmp % p
        elif t != 255:
            acc = acc * tab[t] % p
    assert acc == pow(x, e, p)


# ---- the two-row "fold" layout of csrc/coop2d.h (tools/coop2d_model.py is its executable specification) ------------
FOLD_SLACK = 34      # R'/p >= 2^34: a folded product of inputs < 2^35 p stays < 2^33 p


def fold_w(p):
    """27-bit limbs for the 253..255-bit fields (11 limbs either way, and the 21 products of a result column then
    leave a 32-bit carry, so the last carry pass can start from the 64-bit sum over the rows); 28 for the 377/381-bit
    fields (15 limbs: 27 bits would need 16)"""
    return 27 if p.bit_length() <= 256 else 28


class FoldLayout:
    def __init__(self, p):
        self.W = fold_w(p)
        self.NL = -(-(p.bit_length() + FOLD_SLACK) // self.W)
        assert self.NL <= 15
        self.Q = (self.NL + 1) // 2
        self.OFF = 16 - self.NL
        self.R = 1 << (self.W * self.NL)
        self.M = (1 << self.W) - 1


def fold_layout(p):
    return FoldLayout(p)


def fold_struct(fl, p, L, g, delta, i21, i43):
    """struct Fold of FieldC: constants of the two-row cooperative arithmetic (W = 28, R' = 2^(28 NL), NL chosen so that
    R'/p >= 2^34).  FoldT is the fold table in the kernels' lane order: word [(q * 2 + h) * 16 + l] = limb l of
    C_(2q+h) = 2^(28 (2q+h)) R'^-1 mod p (0 beyond NL)."""
    import math
    W, NL, Q = fl.W, fl.NL, fl.Q
    Rp = fl.R % p
    rinv = pow(fl.R, -1, p)

    def limbsw(v, n=NL):
        assert 0 <= v < (1 << (W * n))
        return [(v >> (W * i)) & fl.M for i in range(n)]

    # g * x: limb-wise while g * (2^28 + 2^6) fits 32 bits, else a product by g R'.  What the S-box subtracts is
    # g * (a product output) or a product output; product outputs are < 2^32.5 p (tests/test_bounds_walk.py walks the
    # kernels' own code with exact bounds -- anemoi-rust_amd/csrc/BOUNDS.md, "two-row fold" -- and
    # tests/test_coop2d_model.py the model), so the subtraction pad is the next power of two above g * 2^32.5 (or above
    # 2^32.5); x and y are settled once per round, after the linear layer.
    scale_g = g * ((1 << W) + 64) < (1 << 32)
    subk = 1 << (math.ceil(math.log2(g) + 32.5) if scale_g else 33)
    q = limbsw(subk * p)
    kp = [q[0] + (1 << W)] + [q[i] + (1 << W) - 1 for i in range(1, NL - 1)] + [q[NL - 1] - 1]
    assert sum(v << (W * i) for i, v in enumerate(kp)) == subk * p and all(0 <= v < (1 << (W + 1)) for v in kp)
    assert (subk + (1 << 36)) * p < fl.R, "pad + operands must stay below R'"
    s = []
    s.append("  struct Fold {  // two-row cooperative layout (coop2d.h): radix 2^%d, %d limbs, R' = 2^%d, R'/p = 2^%.1f" % (
        W, NL, W * NL, W * NL - math.log2(p)))
    s.append("    static constexpr int W = %d, NL = %d, Q = %d, OFF = %d;" % (W, NL, Q, fl.OFF))
    s.append("    static constexpr bool kScaleG = %s;  // g * x limb-wise (else a product by g R')" % ("true" if scale_g else "false"))
    s.append("    static constexpr uint32_t kN0Inv = 0x%08xu;  // -p^-1 mod 2^W (mul_exact only)" % ((-pow(p, -1, 1 << W)) % (1 << W)))
    for nm, v in (("P", p), ("One", Rp), ("RR", Rp * Rp % p),
                  ("In", pow(2, 2 * W * NL - 64 * L, p)), ("Out", pow(2, 64 * L, p)),
                  ("Delta", delta * Rp % p), ("GMont", g * Rp % p)):
        s.append("    static constexpr uint32_t %s[%d] = %s;" % (nm, NL, c32(limbsw(v))))
    s.append("    static constexpr uint32_t KP[%d] = %s;  // 2^%d p, limb-padded (subtraction pad)" % (
        NL, c32(kp), subk.bit_length() - 1))
    tab = []
    for qq in range(Q):
        for hh in range(2):
            k = 2 * qq + hh
            ck = ((1 << (W * k)) * rinv) % p if k < NL else 0
            tab += limbsw(ck) + [0] * (16 - NL)
    s.append("    static constexpr uint32_t FoldT[%d] = %s;" % (len(tab), c32(tab)))
    # the same table for FOUR rows per element (row r takes the limbs k = r mod 4): word [(q * 4 + r) * 16 + l] = limb l of
    # C_(4q+r).  11-limb fields only: the four-row S form shifts an operand up by three lanes (NL + 3 <= 16).
    Q4 = (NL + 3) // 4 if NL <= 13 else 0
    tab4 = []
    for qq in range(Q4):
        for rr in range(4):
            k = 4 * qq + rr
            ck = ((1 << (W * k)) * rinv) % p if k < NL else 0
            tab4 += limbsw(ck) + [0] * (16 - NL)
    s.append("    static constexpr int Q4 = %d;   // steps per phase of the four-row form (0: not built for this field)" % Q4)
    s.append("    static constexpr uint32_t FoldT4[%d] = %s;" % (max(len(tab4), 1), c32(tab4 or [0])))
    for nm, inst in (("21", i21), ("43", i43)):
        for cd in ("c", "d"):
            vals = []
            for v in inst["ark_" + cd]:
                vals += limbsw(int(v) * Rp % p)
            s.append("    static constexpr uint32_t Ark%s_%s[%d] = %s;" % (cd.upper(), nm, len(vals), c32(vals)))
    s.append("  };")
    return s


def main():
    with open(os.path.join(ROOT, "tests", "golden", "params.json")) as f:
        params = json.load(f)

    # ------------------------------------------------------------------ oracle (C)
    o = []
    o.append("/* GENERATED by tools/gen_params.py from tests/golden/params.json -- do not edit.")
    o.append(" * Data only: field moduli and the reference's instance parameters")
    o.append(" * (src/<f>/sbox.rs:7-25, src/<f>/anemoi_X_Y/mod.rs:20-32, round_constants.rs),")
    o.append(" * converted to Montgomery form with R = 2^(64*limbs). */")
    o.append("#ifndef ANEMOI_ORACLE_PARAMS_GEN_H\n#define ANEMOI_ORACLE_PARAMS_GEN_H\n#include <stdint.h>")
    o.append("#define ORC_NFIELDS %d\n#define ORC_MAXL %d" % (len(FIELD_IDS), MAXL))
    o.append("typedef struct {\n  const char *name; int limbs, alpha, g, chunk, rounds21, rounds43;\n"
             "  uint64_t n0inv;\n  uint64_t p[6], one[6], r2[6], delta[6], gmont[6], inv_alpha[6];\n"
             "  uint64_t c21[21*6], d21[21*6], c43[28*6], d43[28*6];\n} orc_field;")
    o.append("static const orc_field ORC_FIELDS[ORC_NFIELDS] = {")

    # ------------------------------------------------------------------ HIP / host (C++)
    h = []
    h.append("// GENERATED by tools/gen_params.py from tests/golden/params.json -- do not edit.")
    h.append("// Field and instance constants for the MI355X kernels (32-bit limbs, Montgomery R = 2^(32*N32)")
    h.append("// = 2^(64*limbs): identical to the arkworks in-memory form, see include/anemoi_mi355x.h).")
    h.append("#pragma once\n#include <cstdint>")
    h.append("// bit mask of field ids whose lane-private kernels run on 30-bit limbs (13 instead of 14 limbs for the")
    h.append("// 381/377-bit fields); 0 keeps them on 29-bit limbs (A/B)")
    h.append("#ifndef ANEMOI_RADIX30_FIELDS\n#define ANEMOI_RADIX30_FIELDS 3\n#endif")
    h.append("namespace anemoi {")
    h.append("template <int FIELD> struct FieldC;")
    h.append("constexpr int kNumFields = %d;" % len(FIELD_IDS))

    for fid, name in enumerate(FIELD_IDS):
        fp = params[name]
        p, L = int(fp["modulus"]), fp["u64_limbs"]
        R = pow(2, 64 * L, p)
        n0inv = (-pow(p, -1, 1 << 64)) % (1 << 64)
        delta, e, g = int(fp["delta"]), int(fp["inv_alpha"]), fp["beta"]
        i21, i43 = fp["instances"]["anemoi_2_1"], fp["instances"]["anemoi_4_3"]

        def mont_table(vals, cap):
            out = []
            for v in vals:
                out += limbs64(int(v) * R % p, L)
            return out + [0] * (cap - len(out))

        o.append('  { "%s", %d, %d, %d, %d, %d, %d, 0x%016xULL,' % (
            name, L, fp["alpha"], g, fp["byte_chunk"], i21["num_rounds"], i43["num_rounds"], n0inv))
        for v in (p, R, R * R % p, delta * R % p, g * R % p, e):
            o.append("    %s," % c64(limbs64(v, MAXL)))
        o.append("    %s," % c64(mont_table(i21["ark_c"], 21 * 6)))
        o.append("    %s," % c64(mont_table(i21["ark_d"], 21 * 6)))
        o.append("    %s," % c64(mont_table(i43["ark_c"], 28 * 6)))
        o.append("    %s }," % c64(mont_table(i43["ark_d"], 28 * 6)))

        N32 = 2 * L
        h.append("template <> struct FieldC<%d> {" % fid)
        h.append('  static constexpr const char* kName = "%s";' % name)
        h.append("  static constexpr int kId = %d, N = %d, L64 = %d, kAlpha = %d, kG = %d, kChunk = %d;" % (
            fid, N32, L, fp["alpha"], g, fp["byte_chunk"]))
        h.append("  static constexpr int kRounds21 = %d, kRounds43 = %d;" % (i21["num_rounds"], i43["num_rounds"]))
        h.append("  static constexpr bool kLazy = %s;  // 4p <= R: Montgomery products of inputs < 2p stay < 2p" % (
            "true" if 4 * p <= (1 << (64 * L)) else "false"))
        h.append("  static constexpr uint32_t kN0Inv = 0x%08xu;  // -p^-1 mod 2^32" % (n0inv & 0xFFFFFFFF))
        for nm, v in (("P", p), ("P2", 2 * p), ("One", R), ("R2", R * R % p), ("Delta", delta * R % p),
                      ("GMont", g * R % p)):
            h.append("  static constexpr uint32_t %s[%d] = %s;" % (nm, N32, c32(limbs32(v, N32))))
        for nm, inst in (("21", i21), ("43", i43)):
            for cd in ("c", "d"):
                vals = []
                for v in inst["ark_" + cd]:
                    vals += limbs32(int(v) * R % p, N32)
                h.append("  static constexpr uint32_t Ark%s%s[%d] = %s;" % (cd.upper(), nm, len(vals), c32(vals)))
        # ---- unsaturated-limb constants (csrc/mont29.h): NL limbs of W bits, R' = 2^(W*NL).
        # R29: every field (the wave-cooperative kernels always use it).  R30: the 381/377-bit fields
        # only -- 13 limbs instead of 14 for the lane-private kernels (`Lane` selects it).
        def radix_struct(W):
            NL = -(-(p.bit_length() + 6) // W)  # at least 6 bits of headroom above p
            Rp = pow(2, W * NL, p)
            H = (1 << (W * NL)) // p
            tight = H < (1 << 14)  # little headroom: keep every value below ~33 p (mont29.h)
            subk = 4 if tight else 64
            if tight and g == 2 and H >= 600:
                # BLS12-381 on 30-bit limbs (H = 630): g*x stays a limb-wise doubling and subtraction
                # pads with 8p -- S-box values < 9.1p, linear layer < 239p < H p (walked: csrc/BOUNDS.md)
                tight, subk = False, 8

            def limbsw(v, n=NL):
                assert v < (1 << (W * n))
                return [(v >> (W * i)) & ((1 << W) - 1) for i in range(n)]

            # K*p with every limb but the top one >= 2^W, so limb-wise (K*p)_i - b_i never goes negative
            q = limbsw(subk * p)
            kp = [q[0] + (1 << W)] + [q[i] + (1 << W) - 1 for i in range(1, NL - 1)] + [q[NL - 1] - 1]
            assert sum(v << (W * i) for i, v in enumerate(kp)) == subk * p and all(0 <= v < (1 << (W + 1)) for v in kp)
            s = []
            s.append("  struct R%d {  // radix 2^%d: %d limbs, Montgomery R' = 2^%d; floor(R'/p) = 2^%.1f" % (
                W, W, NL, W * NL, (W * NL) - __import__("math").log2(p)))
            s.append("    static constexpr int W = %d, NL = %d, kSubK = %d;" % (W, NL, subk))
            s.append("    static constexpr bool kTight = %s;" % ("true" if tight else "false"))
            s.append("    static constexpr double kH = %r;  // R'/p: mont products stay < 2p while A*B <= kH" % float(H))
            s.append("    static constexpr uint32_t kN0Inv = 0x%08xu;  // -p^-1 mod 2^W" % ((-pow(p, -1, 1 << W)) % (1 << W)))
            for nm, v in (("P", p), ("One", Rp), ("RR", Rp * Rp % p),
                          ("In", pow(2, 2 * W * NL - 64 * L, p)), ("Out", pow(2, 64 * L, p)),
                          ("Delta", delta * Rp % p), ("GMont", g * Rp % p)):
                s.append("    static constexpr uint32_t %s[%d] = %s;" % (nm, NL, c32(limbsw(v))))
            s.append("    static constexpr uint32_t KP[%d] = %s;  // kSubK * p, limb-padded" % (NL, c32(kp)))
            for nm, inst in (("21", i21), ("43", i43)):
                for cd in ("c", "d"):
                    vals = []
                    for v in inst["ark_" + cd]:
                        vals += limbsw(int(v) * Rp % p)
                    s.append("    static constexpr uint32_t Ark%s_%s[%d] = %s;" % (cd.upper(), nm, len(vals), c32(vals)))
            s.append("  };")
            return s

        h += radix_struct(29)
        h.append("  using Coop = R29;  // limb layout of the wave-cooperative kernels (coop29.h)")
        h += fold_struct(fold_layout(p), p, L, g, delta, i21, i43)
        if p.bit_length() > 300:
            h += radix_struct(30)
            h.append("#if (ANEMOI_RADIX30_FIELDS >> %d) & 1" % fid)
            h.append("  using Lane = R30;  // limb layout of the lane-private kernels (mont29.h)")
            h.append("#else\n  using Lane = R29;\n#endif")
        else:
            h.append("  using Lane = R29;  // limb layout of the lane-private kernels (mont29.h)")
        uses_tmp, cost = False, {}
        for k in (2, 3, 4, 5):
            first, steps = sliding_window(e, k)
            cost[k] = (sum(s_ for s_, _ in steps) + sum(1 for _, t_ in steps if t_ not in (253, 255)) + (1 << (k - 1)), -k)
            uses_tmp = uses_tmp or any(t_ in (253, 254) for _, t_ in steps)
            check_schedule(e, k, first, steps, p)
            flat = []
            for s_, t_ in steps:
                assert s_ < 256
                flat += [s_, t_]
            nsq = sum(s_ for s_, _ in steps)
            nmul = sum(1 for _, t_ in steps if t_ not in (253, 255))
            h.append("  // window %d: %d squarings + %d multiplications (+ %d to build the table)" % (
                k, nsq, nmul, 1 << (k - 1)))
            h.append("  static constexpr int kW%dFirst = %d, kW%dSteps = %d;" % (k, first, k, len(steps)))
            h.append("  static constexpr uint8_t kW%dSched[%d] = {%s};" % (k, len(flat), ",".join(map(str, flat))))
        h.append("  // cheapest window for the wave-cooperative kernels (a table entry costs them one LDS word per lane)")
        coop_win = min(cost, key=cost.get)
        h.append("  static constexpr int kCoopWin = %d;" % coop_win)
        _, csteps = sliding_window(e, coop_win)
        # A schedule is [prefix of leading-run doubling steps (253 / 254)] + [a regular tail]: every tail step is >= 1
        # squarings and then a table multiplication, except that the last may be squarings only.  The cooperative S-box
        # runs the prefix through its general step loop and the tail through a branch-free one (anemoi_coop_kernels.h).
        prefix = 0
        while prefix < len(csteps) and csteps[prefix][1] in (253, 254):
            prefix += 1
        tail = csteps[prefix:]
        regular = (len(tail) >= 2 and all(s_ >= 1 and t_ < 253 for s_, t_ in tail[:-1]) and tail[-1][0] >= 1 and
                   (tail[-1][1] < 253 or tail[-1][1] == 255))
        h.append("  // the kCoopWin schedule = kCoopPrefix leading-run doubling steps + a tail that is \"regular\" (>= 1 squarings, then a")
        h.append("  // table multiplication; the last step may be squarings only): the cooperative S-box runs the tail branch-free")
        h.append("  static constexpr int kCoopPrefix = %d;" % prefix)
        h.append("  static constexpr bool kCoopRegular = %s;" % ("true" if regular else "false"))
        h.append("  static constexpr bool kChainTmp = %s;  // the schedules use the tmp register (ops 253 / 254)" % (
            "true" if uses_tmp else "false"))
        # window-3 table + extra digits held in VGPRs (anemoi_perm.h exp_inv_alpha); not for the fields whose exponent
        # starts with a long run that the tmp doubling already handles (Pallas / Vesta)
        n_extra = 0 if uses_tmp else XDIGITS.get(name, 0)
        extras, xfirst, xsteps, xcost = extra_digit_schedule(e, n_extra)
        check_extra_schedule(e, extras, xfirst, xsteps, p)
        digits = list(XBASE) + [d for d, _ in extras]
        src_id = dict(XSRC)
        if extras:
            src_id[extras[0][0]] = 5
        ops_flat, args_flat, lens = [], [], []
        for d, ops in extras:
            lens.append(len(ops))
            for op, arg in ops:
                ops_flat.append({"load": 0, "sqr": 1, "mul": 2}[op])
                args_flat.append(arg if op == "sqr" else src_id[arg])
        flat = []
        for s_, d_ in xsteps:
            assert s_ < 256
            flat += [s_, 255 if d_ is None else digits.index(d_)]
        nsq = sum(s_ for s_, _ in xsteps) + sum(a for o, a in zip(ops_flat, args_flat) if o == 1)
        h.append("  // window 3 + %d extra digit(s) %s: %d products in all (table build included) against %d for the plain 3-bit window"
                 % (len(extras), [d for d, _ in extras], xcost + sum(s_ for s_, _ in xsteps), cost[3][0]))
        h.append("  static constexpr int kXDigits = %d, kXFirst = %d, kXSteps = %d;" % (len(extras), digits.index(xfirst), len(xsteps)))
        h.append("  static constexpr int kXDigit[%d] = {%s};" % (max(len(extras), 1), ",".join(str(d) for d, _ in extras) or "0"))
        h.append("  static constexpr int kXProgLen[%d] = {%s};  // build programme of each extra: ops 0 = LOAD src, 1 = SQR k, 2 = MUL src"
                 % (max(len(lens), 1), ",".join(map(str, lens)) or "0"))
        h.append("  static constexpr int kXProgOp[%d] = {%s};" % (max(len(ops_flat), 1), ",".join(map(str, ops_flat)) or "0"))
        h.append("  static constexpr int kXProgArg[%d] = {%s};  // src: 0 = x, 1 = x^2, 2..4 = x^3, x^5, x^7, 5 = first extra"
                 % (max(len(args_flat), 1), ",".join(map(str, args_flat)) or "0"))
        h.append("  static constexpr uint8_t kXSched[%d] = {%s};" % (max(len(flat), 1), ",".join(map(str, flat)) or "0"))
        h.append("};")

    o.append("};\n#endif")
    h.append("}  // namespace anemoi")

    with open(os.path.join(ROOT, "oracle", "anemoi_params_gen.h"), "w") as f:
        f.write("\n".join(o) + "\n")
    dst = os.path.join(ROOT, "anemoi-rust_amd", "csrc")
    os.makedirs(dst, exist_ok=True)
    with open(os.path.join(dst, "field_consts_gen.h"), "w") as f:
        f.write("\n".join(h) + "\n")
    print("wrote oracle/anemoi_params_gen.h and anemoi-rust_amd/csrc/field_consts_gen.h")


if __name__ == "__main__":
    main()
