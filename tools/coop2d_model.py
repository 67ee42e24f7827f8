#!/usr/bin/env python3
"""Lane-level model of the two-row ("2-D") cooperative Montgomery product of csrc/coop2d.h.

One field element = NL limbs of W bits, limb j in lane j of BOTH 16-lane DPP rows of its row pair (two elements per
wavefront: rows 0-1 and rows 2-3).  A product a * b * R'^-1 mod p  (R' = 2^(W NL)) is

  P1   T = a x b, schoolbook, the multiplier limbs split over the two rows by parity (row h takes a_i, i = h mod 2):
       lane l of LO accumulates column l - OFF (OFF = 16 - NL: columns 0..NL-1 sit in lanes OFF..15), lane l of HI
       column NL + l -- the HIGH half lands where the result limb l will live;
  RN1  the low columns are summed over the two rows (v_permlane16_swap) and carried to limbs t_k < 2^W + 2^5; the
       carry out of column NL-1 is added to HI lane 0;
  P2   the low half is folded away with a table instead of two more products:
           T R'^-1 = TH + sum_k t_k C_k   (mod p),   C_k = 2^(W k) R'^-1 mod p,
       C_k[l] a per-lane constant, again split over the rows by parity and accumulated ONTO the HI accumulators;
  RN2  HI is carried per row, summed over the rows and carried again (W = 28), or summed over the rows as 64-bit values
       and carried twice (W = 27): limbs < 2^W + 2^5, value < (AB/H + NL 2^W) p.

No quotient digit, no digit-serial dependency: Q = ceil(NL / 2) multiply-add steps per phase instead of NL.
This file is the executable specification: run() executes the exact lane program of the kernel on Python integers with
32-/64-bit overflow checks, so tests/test_asm_model.py can drive it with adversarial limbs.
"""
import random

ROWS = 2  # rows per element


class Overflow(Exception):
    pass


def chk(v, bits, what):
    if v < 0 or v >> bits:
        raise Overflow("%s needs more than %d bits: %d" % (what, bits, v.bit_length()))
    return v


class Layout:
    def __init__(self, p, W, NL=None, slack=34):
        self.p, self.W = p, W
        self.NL = NL or -(-(p.bit_length() + slack) // W)
        assert self.NL <= 15, "a row has 16 lanes and lane 15 must stay zero for NL = 15"
        self.Q = (self.NL + 1) // 2
        self.OFF = 16 - self.NL
        self.R = 1 << (W * self.NL)
        self.M = (1 << W) - 1
        self.H = self.R // p
        rinv = pow(self.R, -1, p)
        self.C = [((1 << (W * k)) * rinv) % p for k in range(self.NL)]

    def limbs(self, v):
        assert 0 <= v < self.R
        return [(v >> (self.W * i)) & self.M for i in range(self.NL)] + [0] * (16 - self.NL)

    def value(self, l):
        return sum(x << (self.W * i) for i, x in enumerate(l))


# ---- DPP / permlane primitives on one row pair: a register is [row0[16], row1[16]] ----------------------------------
def rep(l):
    return [list(l), list(l)]


def row_shr(reg, n):  # lane l <- lane l - n, zero fill
    return [[r[l - n] if l - n >= 0 else 0 for l in range(16)] for r in reg]


def row_shl(reg, n):  # lane l <- lane l + n, zero fill
    return [[r[l + n] if l + n < 16 else 0 for l in range(16)] for r in reg]


def row_ror(reg, n):  # lane l <- lane (l - n) mod 16
    return [[r[(l - n) % 16] for l in range(16)] for r in reg]


def bcast(reg, n):
    return [[r[n]] * 16 for r in reg]


def odd_rows(old, new):  # row_mask 0xa: only the odd row of the pair is written
    return [old[0], new[1]]


def swap16(vdst, src0):  # v_permlane16_swap: odd rows of vdst <-> even rows of src0
    return [vdst[0], src0[0]], [vdst[1], src0[1]]


def lanes(fn, *regs):
    return [[fn(*(r[h][l] for r in regs)) for l in range(16)] for h in range(2)]


def mul(L, a, b, trace=None):
    """a, b: 16-entry limb lists (plain form).  Returns the 16-entry limb list of the product."""
    W, NL, Q, OFF, M = L.W, L.NL, L.Q, L.OFF, L.M
    A, B = rep(a), rep(b)
    aD = odd_rows(A, row_shl(A, 1))      # row 1: lane l holds a_(l+1)
    bS = odd_rows(B, row_shr(B, 1))      # row 1: lane l holds b_(l-1)
    LO = [[0] * 16, [0] * 16]
    HI = [[0] * 16, [0] * 16]
    for q in range(Q):
        Aq = bcast(aD, 2 * q)
        BL = row_shr(bS, OFF + 2 * q)
        BH = row_shl(bS, NL - 2 * q)
        LO = lanes(lambda t, x, y: chk(t + chk(x, 32, "a limb") * chk(y, 32, "b limb"), 64, "LO"), LO, Aq, BL)
        HI = lanes(lambda t, x, y: chk(t + x * y, 64, "HI"), HI, Aq, BH)
    # RN1: low columns -> limbs t; what leaves the top lane goes to HI lane 0 (column NL)
    if NL <= 13:   # sum over the rows first
        lo = lanes(lambda t: t & M, LO)
        hi = lanes(lambda t: chk(t >> W, 32, "LO >> W"), LO)
        lo, hi = swap16(lo, hi)
        s = lanes(lambda x, y: chk(x + y, 32, "RN1 s"), lo, hi)          # even row: sum of lo, odd row: sum of hi
        X = [list(s[0]), list(s[1])]
        s, X = swap16(s, X)                                               # s = sum lo (both rows), X = sum hi (both rows)
        v = lanes(lambda x, y: chk(x + y, 32, "RN1 v"), s, row_shr(X, 1))
        vh = lanes(lambda x: x >> W, v)
        cc = lanes(lambda x, y: chk(x + y, 32, "RN1 cc"), X, vh)         # lane 15: carry out of column NL-1
        t = lanes(lambda x, y: chk((x & M) + y, 32, "RN1 t"), v, row_shr(vh, 1))
        # row_ror:1, rows 0 / 2 only, bank 0 (lanes 0..3): lane 0 <- cc[15], lanes 1..3 <- cc[0..2] which must be empty
        inj = row_ror(cc, 1)
        assert inj[0][1] == inj[0][2] == inj[0][3] == 0
        ccr = [[inj[0][l] if l < 4 else 0 for l in range(16)], [0] * 16]
    else:          # 15 limbs: carry inside each row first, then sum over the rows
        hi = lanes(lambda t: chk(t >> W, 32, "LO >> W"), LO)
        w = lanes(lambda t, y: chk((t & M) + y, 32, "RN1 w"), LO, row_shr(hi, 1))
        wh = lanes(lambda x: x >> W, w)
        w2 = lanes(lambda x, y: (x & M) + y, w, row_shr(wh, 1))
        cc = lanes(lambda x, y: chk(x + y, 32, "RN1 ccr"), hi, wh)
        ccr = [[cc[h][15] if l == 0 else 0 for l in range(16)] for h in range(2)]   # each row into its own HI lane 0
        X = [list(w2[0]), list(w2[1])]
        w2, X = swap16(w2, X)
        y = lanes(lambda x, z: chk(x + z, 32, "RN1 y"), w2, X)
        yh = lanes(lambda x: x >> W, y)
        t = [[(y[h][l] if l == 15 else y[h][l] & M) + (yh[h][l - 1] if l else 0) for l in range(16)] for h in range(2)]
    tD = odd_rows(t, row_shl(t, 1))
    HI = lanes(lambda x, y: chk(x + y, 64, "HI + cc"), HI, ccr)
    # P2: fold
    for q in range(Q):
        Lq = bcast(tD, OFF + 2 * q)
        CT = [[(L.C[2 * q + h] >> (W * l)) & M if (2 * q + h < NL and l < NL) else 0 for l in range(16)] for h in range(2)]
        HI = lanes(lambda acc, x, y: chk(acc + chk(x, 32, "t limb") * y, 64, "HI fold"), HI, Lq, CT)
    # RN2
    if W <= 27:    # 64-bit sum over the rows first: <= 21 products of 2^54 leave a 32-bit carry
        X = [list(HI[0]), list(HI[1])]
        HIs, X = swap16(HI, X)
        tot = lanes(lambda x, z: chk(x + z, 64, "RN2 total"), HIs, X)
        hi = lanes(lambda x: chk(x >> W, 32, "total >> W"), tot)
        if hi[0][15]:
            raise Overflow("carry out of the top limb")
        w = lanes(lambda x, y: chk((x & M) + y, 32, "RN2 w"), tot, row_shr(hi, 1))
        wh = lanes(lambda x: x >> W, w)
        r = lanes(lambda x, z: (x & M) + z, w, row_shr(wh, 1))
    else:
        lo = lanes(lambda x: x & M, HI)
        hi = lanes(lambda x: chk(x >> W, 32, "HI >> W"), HI)
        w = lanes(lambda x, y: chk(x + y, 32, "RN2 w"), lo, row_shr(hi, 1))
        top = hi[0][15] + hi[1][15]
        if top:
            raise Overflow("carry out of the top limb")
        wh = lanes(lambda x: x >> W, w)
        w2 = lanes(lambda x, y: (x & M) + y, w, row_shr(wh, 1))
        X = [list(w2[0]), list(w2[1])]
        w2, X = swap16(w2, X)
        y = lanes(lambda x, z: chk(x + z, 32, "RN2 y"), w2, X)
        yh = lanes(lambda x: x >> W, y)
        r = lanes(lambda x, z: (x & M) + z, y, row_shr(yh, 1))
    assert r[0] == r[1]
    if trace is not None:
        trace.update(LO=LO, HI=HI, t=t)
    return r[0]


# ---- four rows per element (11-limb fields only: the S form shifts b up by up to three lanes) ---------------------------
def lanes4(fn, *regs):
    return [[fn(*(r[h][l] for r in regs)) for l in range(16)] for h in range(4)]


def swap16_4(vdst, src0):   # v_permlane16_swap on four rows: rows 1 <-> 0' and 3 <-> 2'
    return [vdst[0], src0[0], vdst[2], src0[2]], [vdst[1], src0[1], vdst[3], src0[3]]


def swap32_4(vdst, src0):   # v_permlane32_swap: rows 2, 3 of vdst <-> rows 0, 1 of src0
    return [vdst[0], vdst[1], src0[0], src0[1]], [vdst[2], vdst[3], src0[2], src0[3]]


def shr4(reg, n):
    return [[r[l - n] if l - n >= 0 else 0 for l in range(16)] for r in reg]


def shl4(reg, n):
    return [[r[l + n] if l + n < 16 else 0 for l in range(16)] for r in reg]


def mul4(L, a, b):
    """The four-row form of mul(): one element per wavefront, row r multiplies by the limbs a_i with i = r (mod 4):
    Q4 = ceil(NL / 4) steps per phase.  Same column layout (LO lane l = column l - OFF, HI lane l = column NL + l), the
    sums over the rows take two swap levels (v_permlane16_swap, v_permlane32_swap)."""
    W, NL, OFF, M = L.W, L.NL, L.OFF, L.M
    assert NL <= 13, "row 3 holds b shifted up by three lanes"
    Q4 = (NL + 3) // 4
    A, B = [list(a)] * 4, [list(b)] * 4
    aD = [shl4(A, r)[r] for r in range(4)]       # row r: lane l holds a_(l+r)
    bS = [shr4(B, r)[r] for r in range(4)]       # row r: lane l holds b_(l-r)
    LO = [[0] * 16 for _ in range(4)]
    HI = [[0] * 16 for _ in range(4)]
    for q in range(Q4):
        Aq = [[r[4 * q]] * 16 for r in aD]
        BL, BH = shr4(bS, OFF + 4 * q), shl4(bS, NL - 4 * q)
        LO = lanes4(lambda t, x, y: chk(t + chk(x, 32, "a limb") * chk(y, 32, "b limb"), 64, "LO"), LO, Aq, BL)
        HI = lanes4(lambda t, x, y: chk(t + x * y, 64, "HI"), HI, Aq, BH)
    # RN1: sums over the four rows (cross-first)
    lo = lanes4(lambda t: t & M, LO)
    hi = lanes4(lambda t: chk(t >> W, 32, "LO >> W"), LO)
    lo, hi = swap16_4(lo, hi)
    s = lanes4(lambda x, y: chk(x + y, 32, "RN1 s"), lo, hi)          # rows: lo0+lo1, hi0+hi1, lo2+lo3, hi2+hi3
    x = [list(r) for r in s]
    s, x = swap32_4(s, x)
    u = lanes4(lambda p, q_: chk(p + q_, 32, "RN1 u"), s, x)          # rows: sum lo, sum hi, sum lo, sum hi
    y = [list(r) for r in u]
    u, y = swap16_4(u, y)                                             # u = sum lo, y = sum hi, in all four rows
    v = lanes4(lambda p, q_: chk(p + q_, 32, "RN1 v"), u, shr4(y, 1))
    vh = lanes4(lambda p: p >> W, v)
    cc = lanes4(lambda p, q_: chk(p + q_, 32, "RN1 cc"), y, vh)
    t = lanes4(lambda p, q_: chk((p & M) + q_, 32, "RN1 t"), v, shr4(vh, 1))
    assert cc[0][0] == cc[0][1] == cc[0][2] == 0                      # the injection reads lanes 15, 0, 1, 2 (row_ror:1, bank 0)
    HI[0][0] = chk(HI[0][0] + cc[0][15], 64, "HI + cc")               # row 0 only: the rows are summed in RN2
    tD = [shl4(t, r)[r] for r in range(4)]
    # P2: fold
    for q in range(Q4):
        Lq = [[r[OFF + 4 * q]] * 16 for r in tD]
        CT = [[(L.C[4 * q + h] >> (W * l)) & M if (4 * q + h < NL and l < NL) else 0 for l in range(16)] for h in range(4)]
        HI = lanes4(lambda acc, p, c: chk(acc + chk(p, 32, "t limb") * c, 64, "HI fold"), HI, Lq, CT)
    # RN2: the 64-bit sum over the four rows first
    assert W <= 27
    X = [list(r) for r in HI]
    Hs, X = swap16_4(HI, X)
    two = lanes4(lambda p, q_: chk(p + q_, 64, "RN2 pair sums"), Hs, X)
    X = [list(r) for r in two]
    Hs, X = swap32_4(two, X)
    tot = lanes4(lambda p, q_: chk(p + q_, 64, "RN2 total"), Hs, X)
    hi = lanes4(lambda p: chk(p >> W, 32, "total >> W"), tot)
    if hi[0][15]:
        raise Overflow("carry out of the top limb")
    w = lanes4(lambda p, q_: chk((p & M) + q_, 32, "RN2 w"), tot, shr4(hi, 1))
    wh = lanes4(lambda p: p >> W, w)
    r = lanes4(lambda p, q_: (p & M) + q_, w, shr4(wh, 1))
    assert r[0] == r[1] == r[2] == r[3]
    return r[0]


def selftest(seed=1, rounds=200):
    fields = {
        "bls12_381": 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
        "jubjub": 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
        "bn_254": 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
    }
    rng = random.Random(seed)
    for name, p in fields.items():
        for W in (27, 28):
            if p.bit_length() > 300 and W == 27:
                continue
            L = Layout(p, W)
            bound = min(L.R - 1, (1 << 35) * p)
            worst = 0
            for it in range(rounds):
                if it % 4 == 0:    # adversarial: all limbs at their maximum (2^W + 31), top limb bounded by the value bound
                    a = [L.M + 32] * L.NL
                    b = [L.M + 32] * L.NL
                    topmax = bound >> (W * (L.NL - 1))
                    a[-1] = b[-1] = min(topmax, L.M + 32)
                    a += [0] * (16 - L.NL)
                    b += [0] * (16 - L.NL)
                else:
                    a = L.limbs(rng.randrange(bound))
                    b = L.limbs(rng.randrange(bound))
                r = mul(L, a, b)
                if L.NL <= 13 and W <= 27:
                    assert mul4(L, a, b) == r        # the four-row form gives the same LIMBS, not only the same value
                va, vb, vr = L.value(a), L.value(b), L.value(r)
                assert (vr * L.R - va * vb) % p == 0, (name, W, it)
                assert all(x <= L.M + 32 for x in r[:L.NL]) and all(x == 0 for x in r[L.NL:]), (name, W, r)
                worst = max(worst, vr // p)
            print("%-10s W=%d NL=%d Q=%d H=2^%.1f: %d products ok, largest result %.1f bits above p" % (
                name, W, L.NL, L.Q, __import__("math").log2(L.H), rounds, __import__("math").log2(max(worst, 1))))


if __name__ == "__main__":
    selftest()
