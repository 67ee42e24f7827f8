#!/usr/bin/env python3
"""Generate anemoi-rust_amd/csrc/mont29_asm_gen.h: the lane-private unsaturated-limb Montgomery
squaring and multiplication of mont29.h (29-bit limbs; also 30-bit limbs for the 381/377-bit fields)
as hand-scheduled gfx950 assembly, one `asm volatile` statement per product.

Why: hipcc reassociates the C++ product scanning of mont29.h into an operand-scanning schedule that
needs one extra 64-bit add per column and ~30 register moves per multiplication (412 / 537
instructions per squaring / multiplication on 14 limbs).  The straight product-scanning form below
carries the column carry as the accumulator's initial value: 383 / 474 instructions.  The kernels
are bound by VALU issue (profiles/r01), so instructions are time.

Form (NL limbs, column k of the 2 NL - 1 columns, accumulator ACC = 64-bit VGPR pair):
    ACC += sum a_j * b_{k-j}          v_mad_u64_u32 (squaring: pre-doubled a2_j * a_{k-j}, j < k-j, + a_{k/2}^2)
    ACC += sum m_j * p_{k-j}          v_mad_u64_u32 with p as SGPR operand
    k <  NL:  m_k = (ACC.lo * n0inv) & MASK ; ACC += m_k * p_0
    k >= NL:  out_{k-NL} = ACC.lo & MASK      (written over a_{k-NL}, dead by then)
    ACC >>= W          (+ the parked upper part of a split column, see class Column)
No hazards need padding: only VALU -> VALU register dependences (interlocked), VCC is written (the
unused carry-out of v_mad_u64_u32) but never read.

Operand budget (inline asm allows 30): squaring = NL in/out VGPRs + NL SGPRs (p) + 1 SGPR (n0inv);
multiplication = NL in/out + NL in VGPRs, so p and n0inv are loaded into clobbered SGPRs inside the
statement (NL + 1 s_mov_b32).  Temporaries live in fixed clobbered VGPRs v[TMP_BASE ...].

    python tools/gen_asm_mul.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELD_IDS = ["bls12_381", "bls12_377", "bn_254", "ed_on_bls12_377", "jubjub", "pallas", "vesta"]
TMP_BASE = int(os.environ.get("ANEMOI_ASM_TMP_BASE", "100"))  # clobbered VGPRs start here (even: the 64-bit accumulator is 2-aligned)
TMP_BASE9 = int(os.environ.get("ANEMOI_ASM_TMP_BASE9", str(TMP_BASE)))  # the same for the 9-limb fields (A/B of their occupancy)


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asm_grid import align8  # noqa: E402  (the 8-byte fetch grid: see tools/asm_grid.py)


def tmp_base(nl):
    return TMP_BASE if nl >= 13 else TMP_BASE9
SGPR_BASE = 60      # clobbered SGPRs of the multiplication
CHAINS = int(os.environ.get("ANEMOI_ASM_CHAINS", "1"))   # independent multiply-add chains per column (see Column.mads)
CAP = 1 << 64


def field_consts(p, W):
    nl = -(-(p.bit_length() + 6) // W)
    mask = (1 << W) - 1
    limbs = [(p >> (W * i)) & mask for i in range(nl)]
    n0inv = (-pow(p, -1, 1 << W)) % (1 << W)
    return nl, limbs, n0inv


class Column:
    """Emits the multiply-accumulates of the product columns while tracking a worst-case bound of the
    64-bit accumulator (input limbs < 2^W, reduction digits < 2^W, the actual limbs of p).  With 29-bit
    limbs every column fits.  With 30-bit limbs (13 limbs for 381/377 bits) the five middle columns do
    not: there the accumulator is split once -- its upper word parked in T, the low word kept (the
    reduction digit and the output limb only look at the low W bits) -- and T << (32 - W) is added
    back after the column's shift.

    Column carry.  The carry into the next column is ACC >> W.  A 64-bit shift costs as much as a
    multiply-add (4.3 cycles per wave-instruction, tools/ubench/wall_rates.hip); where the bound shows the
    carry fits 32 bits (ACC < 2^(32 + W): the light columns at both ends -- most columns of a 9-limb
    product) it is taken with ONE 32-bit funnel shift (v_alignbit_b32, 2.4 cycles) into the low half of
    the register pair [C, Z], Z a register that is zero for the whole statement, and the next column's
    first multiply-add reads that pair as its addend (v_mad_u64_u32 ACC, a, b, [C:Z]) -- so the
    accumulator is re-initialised for free.  The first column starts from the inline constant 0."""

    def __init__(self, out, W, acc, treg, cz=None, xacc=()):
        self.out, self.W, self.acc, self.treg, self.cz = out, W, acc, treg, cz
        self.xacc = list(xacc)  # extra accumulator pairs: independent multiply-add chains inside a column
        self.multi = 0          # columns that used them
        self.ACC = "v[%d:%d]" % (acc, acc + 1)
        self.T = "v[%d:%d]" % (treg, treg + 1)
        self.CZ = "v[%d:%d]" % (cz, cz + 1) if cz is not None else None
        self.bound = 0          # upper bound of ACC
        self.tbound = None      # upper bound of T while a split is pending
        self.splits = 0
        self.addend = "0"       # what the next column's first multiply-add adds: "0", ACC or the [C, Z] pair
        self.light = 0          # columns whose carry took the 32-bit path
        if cz is not None:
            out.append("v_mov_b32 v%d, 0" % (cz + 1))

    def mad(self, a, b, amax, bmax):
        term = amax * bmax
        if self.bound + term >= CAP:
            assert self.tbound is None, "a column needs more than one split"
            assert self.addend == self.ACC, "a split before the column's first multiply-add"
            # park the accumulator's upper word: column value = (T << 32) + ACC from here on
            if self.splits == 0:
                self.out.append("v_mov_b32 v%d, 0" % (self.treg + 1))
            self.out.append("v_mov_b32 v%d, v%d" % (self.treg, self.acc + 1))
            self.out.append("v_mov_b32 v%d, 0" % (self.acc + 1))
            self.tbound = self.bound >> 32
            self.bound = (1 << 32) - 1
            self.splits += 1
        self.out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (self.ACC, a, b, self.addend))
        self.addend = self.ACC
        self.bound += term
        assert self.bound < CAP

    def mads(self, terms, reserve=0):
        """The multiply-adds of a column, terms = [(a, b, amax, bmax)].  With extra accumulators the terms go
        round-robin onto 1 + len(xacc) independent chains (a lone wavefront issues dependent and independent
        v_mad_u64_u32 alike, but at 3 waves per SIMD independent chains issue ~2-7 % faster:
        tools/ubench/wall_chains.hip) and are summed with v_lshl_add_u64 at the end -- only in columns whose
        total (plus `reserve`, the m_k p_0 term still to come) provably fits 64 bits, so that no chain and no
        partial sum can overflow; the heavy middle columns of the 30-bit layout keep the single chain and
        its split."""
        total = self.bound + sum(x * y for _, _, x, y in terms)
        n = min(1 + len(self.xacc), len(terms))
        if n < 2 or total + reserve >= CAP or self.tbound is not None:
            for t in terms:
                self.mad(*t)
            return
        started = set()
        for i, (a, b, _, _) in enumerate(terms):
            c = i % n
            if c == 0:
                self.out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (self.ACC, a, b, self.addend))
                self.addend = self.ACC
            else:
                X = "v[%d:%d]" % (self.xacc[c - 1], self.xacc[c - 1] + 1)
                self.out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (X, a, b, X if c in started else "0"))
                started.add(c)
        for c in sorted(started):
            X = "v[%d:%d]" % (self.xacc[c - 1], self.xacc[c - 1] + 1)
            self.out.append("v_lshl_add_u64 %s, %s, 0, %s" % (self.ACC, X, self.ACC))
        self.bound = total
        self.multi += 1

    def low(self):
        """register holding the low word of the column sum (valid once the column has a multiply-add)"""
        assert self.addend == self.ACC
        return "v%d" % self.acc

    def shift(self, final_dst=None):
        """end of a column: ACC >>= W (or the 32-bit carry path).  final_dst: the last column writes its
        (< 2^32) carry straight into the top output limb."""
        assert self.addend == self.ACC
        if self.tbound is None and self.bound < (1 << (32 + self.W)) and (final_dst or self.CZ):
            dst = final_dst or ("v%d" % self.cz)
            self.out.append("v_alignbit_b32 %s, v%d, v%d, %d" % (dst, self.acc + 1, self.acc, self.W))
            self.bound >>= self.W
            self.light += 1
            if not final_dst:
                self.addend = self.CZ
            return
        self.out.append("v_lshrrev_b64 %s, %d, %s" % (self.ACC, self.W, self.ACC))
        self.bound >>= self.W
        if self.tbound is not None:
            self.out.append("v_lshl_add_u64 %s, %s, %d, %s" % (self.ACC, self.T, 32 - self.W, self.ACC))
            self.bound += self.tbound << (32 - self.W)
            assert self.bound < CAP
            self.tbound = None
        if final_dst:
            self.out.append("v_mov_b32 %s, v%d" % (final_dst, self.acc))


def top_limb_bound(p, W, nl, mult):
    """largest top limb of a value < mult * p whose other limbs are normalised (< 2^W)"""
    return min((1 << W) - 1, (mult * p) >> (W * (nl - 1)))


# Preconditions of the 30-bit-limb products (where the column bound is tight), in units of p:
#   sqr(a):    a < 16 p   (S-box values are < 15 p, chain values < 2 p; anemoi_perm.h)
#   mul(a, b): b < 16 p   (b is a table entry or a fresh product < 2 p, a Montgomery-form constant, or the S-box input x < 15 p)
# They shave the top limb's contribution off columns nl-1 .. 2nl-2 and save two of the five splits in
# the squaring, one in the multiplication.  29-bit limbs have slack and assume nothing.
SQR_A_MULT, MUL_B_MULT = 16, 16


def gen_sqr(nl, plimbs, n0inv, W, p=None):
    """operands: %0..%{nl-1} = a (in/out VGPR), %{nl}..%{2nl-1} = p limbs (SGPR), %{2nl} = n0inv (SGPR)"""
    MASK = (1 << W) - 1
    amax = [MASK] * nl
    if W >= 30:
        amax[nl - 1] = top_limb_bound(p, W, nl, SQR_A_MULT)
    TMP_BASE = tmp_base(nl)
    acc = TMP_BASE                 # v[acc:acc+1]
    a2 = TMP_BASE + 2              # nl regs
    m = a2 + nl                    # nl regs
    treg = (m + nl + 1) & ~1      # 2 regs (64-bit aligned), only touched when a column is split
    cz = treg if W == 29 else None   # [carry, zero] pair of the 32-bit carry path (29-bit limbs never split)
    A = lambda i: "%%%d" % i
    P = lambda i: "%%%d" % (nl + i)
    N0 = "%%%d" % (2 * nl)
    out = []
    for j in range(nl - 1):   # a2[nl-1] would never be read: the top limb is always the larger index of a pair
        out.append("v_lshlrev_b32 v%d, 1, %s" % (a2 + j, A(j)))
    xacc = [((treg + 2 + 1) & ~1) + 2 * i for i in range(CHAINS - 1)]
    col = Column(out, W, acc, treg, cz, xacc)
    ACC = col.ACC
    for k in range(2 * nl - 1):
        j0 = 0 if k < nl else k - nl + 1
        j = j0
        terms = []
        while j < k - j:
            terms.append(("v%d" % (a2 + j), A(k - j), 2 * amax[j], amax[k - j]))
            j += 1
        if k % 2 == 0:
            terms.append((A(k // 2), A(k // 2), amax[k // 2], amax[k // 2]))
        if k < nl:
            for j in range(k):
                if plimbs[k - j]:
                    terms.append(("v%d" % (m + j), P(k - j), MASK, plimbs[k - j]))
            col.mads(terms, MASK * plimbs[0])
            if n0inv == MASK:   # p = 1 mod 2^W: m = -lo mod 2^W
                out.append("v_sub_u32 v%d, 0, %s" % (m + k, col.low()))
            else:
                out.append("v_mul_lo_u32 v%d, %s, %s" % (m + k, col.low(), N0))
            out.append("v_and_b32 v%d, 0x%x, v%d" % (m + k, MASK, m + k))
            col.mad("v%d" % (m + k), P(0), MASK, plimbs[0])
        else:
            for j in range(k - nl + 1, nl):
                if plimbs[k - j]:
                    terms.append(("v%d" % (m + j), P(k - j), MASK, plimbs[k - j]))
            col.mads(terms)
            out.append("v_and_b32 %s, 0x%x, %s" % (A(k - nl), MASK, col.low()))   # a_{k-nl} is dead from here on
        col.shift(A(nl - 1) if k == 2 * nl - 2 else None)
    # 13/14-limb fields always claim the full 30-register window (through v129 at the default base): a kernel that ends
    # up with <= 128 VGPRs is allowed 4 wavefronts per SIMD, LDS then caps the CU at 12, and the
    # resulting 4,4,4,0 placement is ~9 % slower than 3,3,3,3 (measured on BLS12-377, 30-bit limbs).
    ntmp = max(treg + 2, TMP_BASE + 30) if nl >= 13 else (treg + 2 if cz is not None else m + nl)
    if xacc:
        ntmp = max(ntmp, xacc[-1] + 2)
    clob = ["v%d" % r for r in range(TMP_BASE, ntmp)] + ["vcc"]
    return out, clob, col.splits, col.light


def gen_mul(nl, plimbs, n0inv, W, p=None):
    """operands: %0..%{nl-1} = a (in/out VGPR), %{nl}..%{2nl-1} = b (VGPR)"""
    MASK = (1 << W) - 1
    bmax = [MASK] * nl
    if W >= 30:
        bmax[nl - 1] = top_limb_bound(p, W, nl, MUL_B_MULT)
    TMP_BASE = tmp_base(nl)
    acc = TMP_BASE
    m = TMP_BASE + 2
    treg = (m + nl + 1) & ~1
    cz = treg if W == 29 else None
    A = lambda i: "%%%d" % i
    B = lambda i: "%%%d" % (nl + i)
    SP = lambda i: "s%d" % (SGPR_BASE + i)
    SN0 = "s%d" % (SGPR_BASE + nl)
    out = []
    consts = [(SP(i), plimbs[i]) for i in range(nl)] + [(SN0, n0inv)]
    # (those with a 32-bit literal first, the 4-byte inline-constant forms together behind them: the 8-byte fetch grid)
    for dst, v in sorted(consts, key=lambda c: c[1] <= 64):
        out.append("s_mov_b32 %s, 0x%x" % (dst, v))
    xacc = [((treg + 2 + 1) & ~1) + 2 * i for i in range(CHAINS - 1)]
    col = Column(out, W, acc, treg, cz, xacc)
    for k in range(2 * nl - 1):
        j0, j1 = (0, k) if k < nl else (k - nl + 1, nl - 1)
        terms = [(A(j), B(k - j), MASK, bmax[k - j]) for j in range(j0, j1 + 1)]
        if k < nl:
            for j in range(k):
                if plimbs[k - j]:
                    terms.append(("v%d" % (m + j), SP(k - j), MASK, plimbs[k - j]))
            col.mads(terms, MASK * plimbs[0])
            if n0inv == MASK:
                out.append("v_sub_u32 v%d, 0, %s" % (m + k, col.low()))
            else:
                out.append("v_mul_lo_u32 v%d, %s, %s" % (m + k, col.low(), SN0))
            out.append("v_and_b32 v%d, 0x%x, v%d" % (m + k, MASK, m + k))
            col.mad("v%d" % (m + k), SP(0), MASK, plimbs[0])
        else:
            for j in range(k - nl + 1, nl):
                if plimbs[k - j]:
                    terms.append(("v%d" % (m + j), SP(k - j), MASK, plimbs[k - j]))
            col.mads(terms)
            out.append("v_and_b32 %s, 0x%x, %s" % (A(k - nl), MASK, col.low()))
        col.shift(A(nl - 1) if k == 2 * nl - 2 else None)
    ntmp = treg + 2 if (col.splits or cz is not None) else m + nl
    if xacc:
        ntmp = max(ntmp, xacc[-1] + 2)
    clob = (["v%d" % r for r in range(TMP_BASE, ntmp)] + ["s%d" % (SGPR_BASE + i) for i in range(nl + 1)] + ["vcc"])
    return out, clob, col.splits, col.light


COOP_VBASE, COOP_SBASE = 60, 60   # fixed (clobbered) registers of the cooperative product


def gen_coop_mul(nl, n0inv, W=29):
    """Wave-cooperative Montgomery product of coop29.h (one W-bit limb per lane, lanes 0..nl-1 of a DPP row)
    as one asm statement.  Operands: %0 = t.lo (out), %1 = t.hi (out), %2 = a, %3 = b, %4 = p limb of this
    lane, %5 = per-lane shift amount (W in lane 0, 63 elsewhere).  Per step i:
        T += a_i * b                       a_i: SGPR, all nl of them broadcast up front with v_readlane
        m  = -(T_lane0.lo) / p mod 2^W     v_readlane -> scalar multiply / mask
        T += m * p_lane
        U  = T >> shift                    lane 0: the retired column's carry; other lanes: 0 (T < 2^63)
        T  = T_{lane+1} + U                v_add_co_u32 / v_addc_co_u32 with DPP row_shl:1 on src0
    hipcc's version of the same step needs 13 issue slots (separate DPP moves, a 64-bit add and an extra
    hazard nop after each broadcast); this one 10.  Hazard slots written out by hand: VALU-written VGPR
    -> v_readlane (1), VALU-written VGPR -> DPP source (2, one of them filled by the shift)."""
    MASK = (1 << W) - 1
    T, U = COOP_VBASE, COOP_VBASE + 2
    TT, UU = "v[%d:%d]" % (T, T + 1), "v[%d:%d]" % (U, U + 1)
    SA = lambda i: "s%d" % (COOP_SBASE + i)
    SM = "s%d" % (COOP_SBASE + nl)
    dpp = "row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
    out = ["s_nop 1"]   # whatever VALU wrote `a` last: settled before the first v_readlane
    for i in range(nl):
        out.append("v_readlane_b32 %s, %%2, %d" % (SA(i), i))
    out.append("s_nop 1")
    for i in range(nl):
        out.append("v_mad_u64_u32 %s, vcc, %s, %%3, %s" % (TT, SA(i), TT if i else "0"))
        out.append("s_nop 0")
        out.append("v_readlane_b32 %s, v%d, 0" % (SM, T))
        if n0inv == MASK:
            out.append("s_sub_i32 %s, 0, %s" % (SM, SM))
        else:
            out.append("s_mul_i32 %s, %s, 0x%x" % (SM, SM, n0inv))
        out.append("s_and_b32 %s, %s, 0x%x" % (SM, SM, MASK))
        out.append("v_mad_u64_u32 %s, vcc, %s, %%4, %s" % (TT, SM, TT))
        out.append("v_lshrrev_b64 %s, %%5, %s" % (UU, TT))
        out.append("s_nop 0")
        out.append("v_add_co_u32_dpp v%d, vcc, v%d, v%d %s" % (T, T, U, dpp))
        out.append("v_addc_co_u32_dpp v%d, vcc, v%d, v%d, vcc %s" % (T + 1, T + 1, U + 1, dpp))
    out.append("v_mov_b32 %%0, v%d" % T)
    out.append("v_mov_b32 %%1, v%d" % (T + 1))
    out.append("s_nop 1")  # the caller's next instructions read the results through DPP
    clob = (["v%d" % r for r in range(COOP_VBASE, COOP_VBASE + 4)] +
            ["s%d" % (COOP_SBASE + i) for i in range(nl + 1)] + ["vcc"])
    return out, clob


def gen_coop4_mul(nl, n0inv, W=29):
    """Row-cooperative Montgomery product: FOUR elements per wavefront, one per 16-lane DPP row (lane 16 r + j holds
    limb j of element r, j < nl <= 14; lanes j >= nl hold zero) -- the same operand scan as gen_coop_mul(), but nothing
    in it may be wave-uniform any more: a_i and the quotient digit differ from row to row, so
        a_i   is broadcast inside each row by DPP (v_mov_b32_dpp row_newbcast:i) instead of v_readlane -> SGPR,
        m     = -(T_j=0.lo) / p mod 2^W  is computed on the VALU in every lane and lane 0's value broadcast and masked
                by ONE VOP2-DPP instruction (v_and_b32_dpp m, x, MASK row_newbcast:0) instead of readlane + SALU.
    Operands: %0 = t.lo (out), %1 = t.hi (out), %2 = a, %3 = b, %4 = p limb of this lane, %5 = per-lane shift amount
    (W in lane j = 0 of each row, 63 elsewhere), %6 = MASK (a VGPR: src1 of a VOP2).  Per step i:
        T += a_i * b
        x  = T.lo * n0inv                    (v_sub_u32 x, 0, T.lo when p = 1 mod 2^W)
        m  = bcast_row(x, 0) & MASK          (two wait states in front of this DPP read)
        T += m * p_lane
        U  = T >> shift                      lane 0 of a row: the retired column's carry; other lanes: 0 (T < 2^63)
                                             (the next a_i is broadcast in the wait state behind the shift)
        T  = T_{lane+1} + U                  v_add_co_u32 / v_addc_co_u32 with DPP row_shl:1 on src0
    10 issue slots per step like the one-element scan (whose quotient digit runs on the scalar ALU).  Hazard slots
    by hand: VALU-written VGPR -> DPP source needs 2 wait states."""
    MASK = (1 << W) - 1
    T, U, X, M, A0 = COOP_VBASE, COOP_VBASE + 2, COOP_VBASE + 4, COOP_VBASE + 5, COOP_VBASE + 6
    TT, UU = "v[%d:%d]" % (T, T + 1), "v[%d:%d]" % (U, U + 1)
    SN = "s%d" % COOP_SBASE
    shl = "row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
    bc = lambda i: "row_newbcast:%d row_mask:0xf bank_mask:0xf" % i
    A = lambda i: "v%d" % (A0 + (i & 1))
    out = ["s_nop 1"]   # whatever VALU wrote `a` last: settled before the first DPP read
    if n0inv != MASK:
        out.append("s_mov_b32 %s, 0x%x" % (SN, n0inv))
    out.append("v_mov_b32_dpp %s, %%2 %s" % (A(0), bc(0)))
    for i in range(nl):
        out.append("v_mad_u64_u32 %s, vcc, %s, %%3, %s" % (TT, A(i), TT if i else "0"))
        if n0inv == MASK:
            out.append("v_sub_u32 v%d, 0, v%d" % (X, T))
        else:
            out.append("v_mul_lo_u32 v%d, v%d, %s" % (X, T, SN))
        out.append("s_nop 1")       # (two wait states as ONE instruction: tools/asm_grid.py may write it as two s_nop 0 = 8 bytes)
        out.append("v_and_b32_dpp v%d, v%d, %%6 %s" % (M, X, bc(0)))
        out.append("v_mad_u64_u32 %s, vcc, v%d, %%4, %s" % (TT, M, TT))
        out.append("v_lshrrev_b64 %s, %%5, %s" % (UU, TT))
        if i + 1 < nl:
            out.append("v_mov_b32_dpp %s, %%2 %s" % (A(i + 1), bc(i + 1)))   # the next a_i fills the wait state (8 bytes: the
        else:                                                                 # step stays on the 8-byte fetch grid)
            out.append("v_mov_b32_dpp %s, %%2 %s" % (A(i + 1), bc(0)))       # (last step: a broadcast nobody reads, for the same 8 bytes)
        out.append("v_add_co_u32_dpp v%d, vcc, v%d, v%d %s" % (T, T, U, shl))
        out.append("v_addc_co_u32_dpp v%d, vcc, v%d, v%d, vcc %s" % (T + 1, T + 1, U + 1, shl))
    out.append("v_mov_b32 %%0, v%d" % T)
    out.append("v_mov_b32 %%1, v%d" % (T + 1))
    out.append("s_nop 1")  # the caller's next instructions read the results through DPP
    clob = ["v%d" % r for r in range(COOP_VBASE, A0 + 2)] + ([SN] if n0inv != MASK else []) + ["vcc"]
    return out, clob


def coop_qp_params(p, W=29):
    """Quotient-pipelined form of the cooperative product (Orup 1995, delay d = 1), radix 2^W:
    Mt = (-p^-1 mod 2^2W) p  (= -1 mod 2^2W),  Mq = (Mt + 1) / 2^2W,  n = number of W-bit digits with 4 Mt < 2^(W n)
    raised until the limb-padded subtraction constant (all n - 1 low limbs >= 2^W) leaves room for the S-box's
    operand growth.  A product of A, B <= 2 Mt is  = A B 2^(-W n) mod p  and  <= 2 Mt."""
    Mp = (-pow(p, -1, 1 << (2 * W))) % (1 << (2 * W))
    Mt = Mp * p
    assert (Mt + 1) % (1 << (2 * W)) == 0
    Mq = (Mt + 1) >> (2 * W)
    n = 1
    while (1 << (W * n)) <= 4 * Mt:
        n += 1
    # head room H' = 2^(W n) / (4 Mt): products stay <= 2 Mt while (A / 2Mt)(B / 2Mt) <= H'.  Subtraction pads with
    # a multiple of p whose n - 1 low limbs are all >= 2^W - 1, i.e. ~2^(W n) / 2^W ... 2^(W n): one more digit
    # when that would eat the head room (the 253..255-bit fields: n = 11 -> 12)
    if (1 << (W * n)) // (4 * Mt) < (1 << 12):
        n += 1
    return Mt, Mq, n


def gen_coop_mul_qp(n, W=29):
    """Quotient-pipelined cooperative product, one asm statement.  Lane j < n holds limb j.
    Operands: %0 = t.lo (out), %1 = t.hi (out), %2 = a, %3 = b, %4 = Mq limb of this lane, %5 = Mq limb of lane - 1
    (Mq shifted up one lane), %6 = per-lane shift amount (W in lane 0, 63 elsewhere).
        S_1 = a_0 B
        for i = 1 .. n:   q_i = S_i mod 2^W ;  S_{i+1} = (S_i >> W) + a_i B + q_{i-1} Mq      (a_n = 0, q_0 = 0)
        result = S_{n+1} + 2^W q_n Mq          (= 2^W S_{n+2} + q_{n+1} of the textbook form)
    The quotient digit q_i is read one step before it is used, so it is off the dependent chain: per step
    shift -> lane-shift add (2) -> multiply-add (2), with v_readlane / s_and in the shadow -- against
    multiply-add -> readlane -> s_mul -> s_and -> multiply-add -> shift -> add (2) of gen_coop_mul."""
    MASK = (1 << W) - 1
    T, U = COOP_VBASE, COOP_VBASE + 2
    TT, UU = "v[%d:%d]" % (T, T + 1), "v[%d:%d]" % (U, U + 1)
    SA = lambda i: "s%d" % (COOP_SBASE + i)
    SQ = lambda i: "s%d" % (COOP_SBASE + n + (i & 1))
    dpp = "row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
    out = ["s_nop 1"]
    for i in range(n):
        out.append("v_readlane_b32 %s, %%2, %d" % (SA(i), i))
    out.append("s_nop 1")
    out.append("v_mad_u64_u32 %s, vcc, %s, %%3, 0" % (TT, SA(0)))
    for i in range(1, n + 1):
        out.append("v_lshrrev_b64 %s, %%6, %s" % (UU, TT))
        out.append("v_readlane_b32 %s, v%d, 0" % (SQ(i), T))
        out.append("s_and_b32 %s, %s, 0x%x" % (SQ(i), SQ(i), MASK))
        out.append("v_add_co_u32_dpp v%d, vcc, v%d, v%d %s" % (T, T, U, dpp))
        out.append("v_addc_co_u32_dpp v%d, vcc, v%d, v%d, vcc %s" % (T + 1, T + 1, U + 1, dpp))
        if i < n:
            out.append("v_mad_u64_u32 %s, vcc, %s, %%3, %s" % (TT, SA(i), TT))
        if i >= 2:
            out.append("v_mad_u64_u32 %s, vcc, %s, %%4, %s" % (TT, SQ(i - 1), TT))
    out.append("v_mad_u64_u32 %s, vcc, %s, %%5, %s" % (TT, SQ(n), TT))
    out.append("v_mov_b32 %%0, v%d" % T)
    out.append("v_mov_b32 %%1, v%d" % (T + 1))
    out.append("s_nop 1")
    clob = (["v%d" % r for r in range(COOP_VBASE, COOP_VBASE + 4)] +
            ["s%d" % (COOP_SBASE + i) for i in range(n + 2)] + ["vcc"])
    return out, clob


def emit(name, lines, outs, ins, clob):
    body = "\n".join('        "%s\\n\\t"' % l for l in lines)
    return ("    asm volatile(\n%s\n        : %s\n        : %s\n        : %s);\n"
            % (body, ", ".join(outs), ", ".join(ins) if ins else "", ", ".join('"%s"' % c for c in clob)))


def main():
    with open(os.path.join(ROOT, "tests", "golden", "params.json")) as f:
        params = json.load(f)
    h = ["// GENERATED by tools/gen_asm_mul.py -- do not edit.  Hand-scheduled gfx950 Montgomery squaring /",
         "// multiplication on unsaturated limbs (product scanning, one v_mad_u64_u32 per limb product, the",
         "// column carry is the accumulator's initial value).  See the generator's docstring.",
         "#pragma once", "#include <hip/hip_runtime.h>", "#include <cstdint>", "namespace anemoi {",
         "template <int FIELD, int W> struct AsmMont;"]
    def macro(name, lines):
        """a multi-line string-literal macro holding an asm body"""
        return "#define %s \\\n" % name + " \\\n".join('        "%s\\n\\t"' % l for l in lines)

    pinned = []   # (fid, W, nl) of every layout, for the dispatcher macros below
    for fid, name in enumerate(FIELD_IDS):
        p = int(params[name]["modulus"])
        for W in ((29, 30) if p.bit_length() > 300 else (29,)):
            nl, pl, n0 = field_consts(p, W)
            sq, sq_clob, sq_splits, sq_light = gen_sqr(nl, pl, n0, W, p)
            mu, mu_clob, mu_splits, mu_light = gen_mul(nl, pl, n0, W, p)
            (sq, sq_pad), (mu, mu_pad) = align8(sq), align8(mu)
            assert sq_pad + mu_pad == 0, (name, W, sq_pad, mu_pad)   # (8-byte instructions left off the grid)
            nmad = sum(1 for l in sq if l.startswith("v_mad"))
            tag = "%d_%d" % (fid, W)
            h.append("// %s, %d-bit limbs: %d limbs; squaring %d instructions (%d v_mad_u64_u32, %d split columns), "
                     "multiplication %d (%d, %d)" % (name, W, nl, len(sq), nmad, sq_splits, len(mu),
                                                     sum(1 for l in mu if l.startswith("v_mad")), mu_splits))
            h.append("//   32-bit column carries (v_alignbit_b32 instead of v_lshrrev_b64): %d of %d columns (squaring), "
                     "%d (multiplication)" % (sq_light, 2 * nl - 1, mu_light))
            # the bodies as string macros: used by the struct's functions (operands = array elements) and by the
            # ANEMOI_PIN_* macros (operands = local register variables pinned to fixed VGPRs, anemoi_perm.h)
            h.append(macro("ANEMOI_ASM_SQR_BODY_" + tag, sq))
            h.append("#define ANEMOI_ASM_SQR_INS_%s %s" % (tag, ", ".join(['"s"(0x%xu)' % v for v in pl] + ['"s"(0x%xu)' % n0])))
            h.append("#define ANEMOI_ASM_SQR_CLOB_%s %s" % (tag, ", ".join('"%s"' % c for c in sq_clob)))
            h.append(macro("ANEMOI_ASM_MUL_BODY_" + tag, mu))
            h.append("#define ANEMOI_ASM_MUL_CLOB_%s %s" % (tag, ", ".join('"%s"' % c for c in mu_clob)))
            A = ["A%d" % i for i in range(nl)]
            h.append("#define ANEMOI_PIN_SQR_%s(%s) asm volatile(ANEMOI_ASM_SQR_BODY_%s : %s : ANEMOI_ASM_SQR_INS_%s : ANEMOI_ASM_SQR_CLOB_%s)"
                     % (tag, ", ".join(A), tag, ", ".join('"+v"(%s)' % a for a in A), tag, tag))
            h.append("#define ANEMOI_PIN_MUL_%s(%s, B) asm volatile(ANEMOI_ASM_MUL_BODY_%s : %s : %s : ANEMOI_ASM_MUL_CLOB_%s)"
                     % (tag, ", ".join(A), tag, ", ".join('"+v"(%s)' % a for a in A),
                        ", ".join('"v"((B)[%d])' % i for i in range(nl)), tag))
            h.append("template <> struct AsmMont<%d, %d> {" % (fid, W))
            h.append("  static constexpr int NL = %d;" % nl)
            h.append("  __device__ static __forceinline__ void sqr(uint32_t (&a)[NL]) {")
            h.append("    ANEMOI_PIN_SQR_%s(%s);" % (tag, ", ".join("a[%d]" % i for i in range(nl))))
            h.append("  }")
            h.append("  __device__ static __forceinline__ void mul(uint32_t (&a)[NL], const uint32_t (&b)[NL]) {")
            h.append("    ANEMOI_PIN_MUL_%s(%s, b);" % (tag, ", ".join("a[%d]" % i for i in range(nl))))
            h.append("  }")
            h.append("};")
            pinned.append((fid, W, nl))
    # dispatchers: ANEMOI_PIN_SQR(FID, W, q) / ANEMOI_PIN_MUL(FID, W, q, B) on the variables q0 .. q13 of the calling scope
    # (FID, W compile-time constants of a template; the branches not taken are discarded)
    def chain(kind, extra):
        out = []
        for k, (fid, W, nl) in enumerate(pinned):
            args = ", ".join("Q##%d" % i for i in range(nl)) + extra
            out.append("  %sif constexpr ((FID) == %d && (W_) == %d) { ANEMOI_PIN_%s_%d_%d(%s); }"
                       % ("else " if k else "", fid, W, kind, fid, W, args))
        return " \\\n".join(out)
    h.append("#define ANEMOI_PIN_SQR(FID, W_, Q) \\\n" + chain("SQR", ""))
    h.append("#define ANEMOI_PIN_MUL(FID, W_, Q, B) \\\n" + chain("MUL", ", B"))
    h.append("// Wave-cooperative product (coop29.h), always on 29-bit limbs: see gen_coop_mul() in the generator.")
    h.append("template <int FIELD> struct AsmCoop;")
    for fid, name in enumerate(FIELD_IDS):
        p = int(params[name]["modulus"])
        nl, pl, n0 = field_consts(p, 29)
        co, co_clob = gen_coop_mul(nl, n0)
        co, off_grid = align8(co)
        h.append("// %s: %d steps, %d instructions (%d of the 8-byte ones off the fetch grid)" % (name, nl, len(co), off_grid))
        h.append("template <> struct AsmCoop<%d> {" % fid)
        h.append("  __device__ static __forceinline__ uint64_t mul(uint32_t a, uint32_t b, uint32_t pl, uint32_t sh) {")
        h.append("    uint32_t lo, hi;")
        h.append(emit("coop", co, ['"=&v"(lo)', '"=&v"(hi)'], ['"v"(a)', '"v"(b)', '"v"(pl)', '"v"(sh)'], co_clob))
        h.append("    return ((uint64_t)hi << 32) | lo;")
        h.append("  }")
        h.append("};")
    h.append("// Row-cooperative product: four elements per wavefront, one per 16-lane DPP row -- see gen_coop4_mul().")
    h.append("template <int FIELD> struct AsmCoop4;")
    for fid, name in enumerate(FIELD_IDS):
        p = int(params[name]["modulus"])
        nl, pl, n0 = field_consts(p, 29)
        co, co_clob = gen_coop4_mul(nl, n0)
        co, off_grid = align8(co)
        h.append("// %s: %d steps, %d instructions (%d of the 8-byte ones off the fetch grid)" % (name, nl, len(co), off_grid))
        h.append("template <> struct AsmCoop4<%d> {" % fid)
        h.append("  __device__ static __forceinline__ uint64_t mul(uint32_t a, uint32_t b, uint32_t pl, uint32_t sh, uint32_t mask) {")
        h.append("    uint32_t lo, hi;")
        h.append(emit("coop4", co, ['"=&v"(lo)', '"=&v"(hi)'],
                      ['"v"(a)', '"v"(b)', '"v"(pl)', '"v"(sh)', '"v"(mask)'], co_clob))
        h.append("    return ((uint64_t)hi << 32) | lo;")
        h.append("  }")
        h.append("};")
    # Quotient-pipelined cooperative product (Orup, delay 1; gen_coop_mul_qp): measured and NOT adopted -- a lone
    # wavefront issues one instruction per ~7 cycles whatever the dependences, so a product's latency is its
    # instruction count, and the pipelined scan needs 2-3 more steps (profiles/r02/ubench_coop_scan_latency.txt:
    # BLS12-381 384 -> 364 ns per product, Jubjub 255 -> 279 ns).  ANEMOI_GEN_COOP_QP=1 emits it for
    # tools/ubench/coop_scan_latency.hip.
    h.append("template <int FIELD> struct AsmCoopQP;")
    for fid, name in enumerate(FIELD_IDS if os.environ.get("ANEMOI_GEN_COOP_QP") == "1" else []):
        p = int(params[name]["modulus"])
        Mt, Mq, n = coop_qp_params(p)
        co, co_clob = gen_coop_mul_qp(n)
        co, off_grid = align8(co)
        h.append("// %s: %d limbs / steps, %d instructions" % (name, n, len(co)))
        h.append("template <> struct AsmCoopQP<%d> {" % fid)
        h.append("  static constexpr int NX = %d;" % n)
        h.append("  __device__ static __forceinline__ uint64_t mul(uint32_t a, uint32_t b, uint32_t mq, uint32_t mqup, uint32_t sh) {")
        h.append("    uint32_t lo, hi;")
        h.append(emit("coopqp", co, ['"=&v"(lo)', '"=&v"(hi)'],
                      ['"v"(a)', '"v"(b)', '"v"(mq)', '"v"(mqup)', '"v"(sh)'], co_clob))
        h.append("    return ((uint64_t)hi << 32) | lo;")
        h.append("  }")
        h.append("};")
    h.append("}  // namespace anemoi")
    dst = os.path.join(ROOT, "anemoi-rust_amd", "csrc", "mont29_asm_gen.h")
    with open(dst, "w") as f:
        f.write("\n".join(h) + "\n")
    print("wrote", dst)


if __name__ == "__main__":
    main()
