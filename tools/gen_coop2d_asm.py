#!/usr/bin/env python3
"""Generate anemoi-rust_amd/csrc/coop2d_asm_gen.h: the two-row fold product of coop2d.h as hand-scheduled gfx950
assembly -- one statement for a multiplication, one for a RUN of squarings (the loop is inside the statement).

Why: a lone wavefront issues one instruction (or s_nop) per ~4.9 cycles, so a product costs its issue slots.  hipcc's
version of Coop2d::mul is 94 slots per squaring on 11 limbs: 76 instructions + 18 hazard s_nop (it re-uses a handful of
registers, so nearly every DPP / permlane read waits for the write before it).  Scheduled by hand the hazard slots are
filled with work that is there anyway -- the HI multiply-adds of P1 are issued inside the carry chain of RN1, loop
control sits in front of the first DPP reads -- and the statement below is 78 / 101 slots (11 / 15 limbs).

The generator is layout-driven (NL, W): nothing depends on the modulus -- the fold table arrives as operands.
`Prog` keeps the issue order, knows which reads are hazard-sensitive (DPP sources, v_permlane16_swap operands: a VGPR
written by a VALU instruction may be read that way only two issue slots later) and pads with s_nop where the order
leaves a gap; tests/test_coop2d_model.py executes the emitted text on a 64-lane interpreter against the big-integer
result, with overflow checks, and re-checks the hazard distances independently.

    python tools/gen_coop2d_asm.py
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VBASE = 80          # clobbered VGPRs v[VBASE ...] (even: 64-bit pairs are 2-aligned)
LAYOUTS = [(11, 27), (15, 28)]   # (limbs, limb bits) of the fold layouts in field_consts_gen.h


class Prog:
    """Issue-ordered instruction list with automatic hazard padding."""

    def __init__(self):
        self.lines = []          # text
        self.slot = 0
        self.wrote = {}          # vgpr name -> slot of its last VALU write
        self.nops = 0
        self.next_v = VBASE
        self.names = {}
        self.pending = []        # deferred independent instructions (text, writes): issued where a hazard gap opens

    def reg(self, name, pair=False):
        if name not in self.names:
            if pair and self.next_v % 2:
                self.next_v += 1
            self.names[name] = self.next_v
            self.next_v += 2 if pair else 1
        n = self.names[name]
        return ("v[%d:%d]" % (n, n + 1)) if pair else "v%d" % n

    def lo(self, name):
        return "v%d" % self.names[name]

    def hi(self, name):
        return "v%d" % (self.names[name] + 1)

    @staticmethod
    def _vregs(tok):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return ["v%d" % i for i in range(int(m.group(1)), int(m.group(2)) + 1)]
        return [tok] if re.fullmatch(r"v\d+|%\d+", tok) else []

    def emit(self, text, writes=(), sensitive=(), salu=False):
        """`sensitive`: operands read through DPP or by a permlane swap (need 2 slots after their last VALU write).
        The destination of a DPP instruction counts as read that way too (hipcc pads it like a source: the lanes a
        row / bank mask disables keep their old value)."""
        if "_dpp " in text:
            sensitive = list(sensitive) + list(writes)

        def gap_needed():
            need = 0
            for tok in sensitive:
                for r in self._vregs(tok):
                    if r in self.wrote:
                        gap = self.slot - self.wrote[r] - 1      # instructions issued in between
                        need = max(need, 2 - gap)
            return need

        need = gap_needed()
        while need > 0 and self.pending:     # a deferred instruction is a better filler than s_nop
            ftext, fwrites = self.pending.pop(0)
            self.lines.append(ftext)
            for tok in fwrites:
                for r in self._vregs(tok):
                    self.wrote[r] = self.slot
            self.slot += 1
            need = gap_needed()
        if need > 0:
            self.lines.append("s_nop %d" % (need - 1))
            self.slot += need
            self.nops += need
        self.lines.append(text)
        if not salu:
            for tok in writes:
                for r in self._vregs(tok):
                    self.wrote[r] = self.slot
        self.slot += 1

    def defer(self, text, writes=()):
        self.pending.append((text, list(writes)))

    def flush(self):
        while self.pending:
            ftext, fwrites = self.pending.pop(0)
            self.emit(ftext, writes=fwrites)

    def barrier_all(self):
        """everything written so far counts as written in the previous slot (loop back-edge / statement entry)"""
        for r in list(self.wrote):
            self.wrote[r] = self.slot - 1


DPP_ALL = "row_mask:0xf bank_mask:0xf bound_ctrl:1"
DPP_ODD = "row_mask:0xa bank_mask:0xf bound_ctrl:1"


def operand_names(nl, square, loop):
    """GCC operand numbers of the statement: outputs first (%0 = a in/out, then the SGPR run length of a squaring
    run), then the inputs: [b], the Q fold-table registers, [for 15 limbs: the limb mask with lane 15 all ones, and
    the mask that is all ones in lane 15 only]."""
    Q = (nl + 1) // 2
    names, k = {"A": "%0"}, 1
    if loop:
        names["CNT"], k = "%1", 2
    if not square:
        names["B"], k = "%%%d" % k, k + 1
    names["CT"] = ["%%%d" % (k + q) for q in range(Q)]
    k += Q
    if nl > 13:
        names["MTOP"], names["ONLY15"] = "%%%d" % k, "%%%d" % (k + 1)
    return names


def gen_product(nl, W, square, loop):
    """The product statement (operands: operand_names()).  Returns (lines, clobbers, info)."""
    Q, OFF = (nl + 1) // 2, 16 - nl
    cross_first = nl <= 13          # RN1: sum over the rows first (sum of the high parts <= NL 2^W fits 32 bits)
    total_first = W <= 27           # RN2: 64-bit sum over the rows first (21 products of 2^54 leave a 32-bit carry)
    P = Prog()
    ops = operand_names(nl, square, loop)
    A, B, CT, CNT = ops["A"], ops.get("B"), ops["CT"], ops.get("CNT")
    MTOP, ONLY15 = ops.get("MTOP"), ops.get("ONLY15")
    MASK = "0x%x" % ((1 << W) - 1)   # a literal costs no issue slot
    R = P.reg
    # the injection pair: [carry | 0]; only lanes 0..3 of the even rows are ever rewritten, the rest stays 0
    INJ = R("inj", pair=True)
    P.emit("v_mov_b32 %s, 0" % P.lo("inj"), writes=[P.lo("inj")])
    P.emit("v_mov_b32 %s, 0" % P.hi("inj"), writes=[P.hi("inj")])
    P.wrote[A] = P.slot - 1        # whatever wrote the operands last: assume the slot before the statement
    if B:
        P.wrote[B] = P.slot - 1
    if loop:
        P.emit("1:", salu=True)
        P.slot -= 1                 # a label is not an issue slot
        P.barrier_all()
    body_start = P.slot
    # ---- P1 ---------------------------------------------------------------------------------------------------------
    AD, BS = R("aD"), (R("bS") if not square else A)
    P.emit("v_mov_b32 %s, %s" % (AD, A), writes=[AD])
    if loop:
        P.emit("s_sub_u32 %s, %s, 1" % (CNT, CNT), salu=True)
    if not square:
        P.emit("v_mov_b32 %s, %s" % (BS, B), writes=[BS])
    P.emit("v_mov_b32_dpp %s, %s row_shl:1 %s" % (AD, A, DPP_ODD), writes=[AD], sensitive=[A])
    P.emit("v_mov_b32_dpp %s, %s row_shr:1 %s" % (BS, B or A, DPP_ODD), writes=[BS], sensitive=[B or A])
    if loop:
        P.emit("s_cmp_lg_u32 %s, 0" % CNT, salu=True)
    LO, HI = R("LO", pair=True), R("HI", pair=True)
    for q in range(Q):
        aq, bl, bh = R("a%d" % q), R("bl%d" % (q & 1)), R("bh%d" % q)
        P.emit("v_mov_b32_dpp %s, %s row_newbcast:%d %s" % (aq, AD, 2 * q, DPP_ALL), writes=[aq], sensitive=[AD])
        P.emit("v_mov_b32_dpp %s, %s row_shr:%d %s" % (bl, BS, OFF + 2 * q, DPP_ALL), writes=[bl], sensitive=[BS])
        assert 1 <= nl - 2 * q <= 15
        P.emit("v_mov_b32_dpp %s, %s row_shl:%d %s" % (bh, BS, nl - 2 * q, DPP_ALL), writes=[bh], sensitive=[BS])
        P.emit("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (LO, aq, bl, LO if q else "0"), writes=[LO])
        # the HI multiply-adds are deferred: they fill the hazard gaps of RN1's carry chain
        P.defer("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (HI, aq, bh, HI if q else "0"), writes=[HI])

    # ---- RN1: low columns -> limbs t (D form); the carry out of column NL-1 -> HI lane 0 ----------------------------
    lo, hi, x = R("lo"), R("hi"), R("x")
    vh, vl, cc, t = R("vh"), R("vl"), R("cc"), R("t")
    P.emit("v_alignbit_b32 %s, %s, %s, %d" % (hi, P.hi("LO"), P.lo("LO"), W), writes=[hi])
    P.emit("v_and_b32 %s, %s, %s" % (lo, MASK, P.lo("LO")), writes=[lo])
    if cross_first:
        P.emit("v_permlane16_swap_b32 %s, %s" % (lo, hi), writes=[lo, hi], sensitive=[lo, hi])
        P.emit("v_add_u32 %s, %s, %s" % (lo, lo, hi), writes=[lo])           # even row: sum of lo; odd row: sum of hi
        P.emit("v_mov_b32 %s, %s" % (x, lo), writes=[x])
        P.emit("v_permlane16_swap_b32 %s, %s" % (lo, x), writes=[lo, x], sensitive=[lo, x])   # lo = low parts, x = high parts
        P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (hi, x, lo, DPP_ALL), writes=[hi], sensitive=[x])   # v
        P.emit("v_lshrrev_b32 %s, %d, %s" % (vh, W, hi), writes=[vh])
        P.emit("v_add_u32 %s, %s, %s" % (cc, x, vh), writes=[cc])             # lane 15: all that column NL-1 hands on
        P.emit("v_and_b32 %s, %s, %s" % (vl, MASK, hi), writes=[vl])
        # t in D form: even rows t_l = vl_l + vh_(l-1); odd rows t_(l+1) = vl_(l+1) + vh_l
        P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 row_mask:0x5 bank_mask:0xf bound_ctrl:1" % (t, vh, vl), writes=[t], sensitive=[vh])
        P.emit("v_mov_b32_dpp %s, %s row_ror:1 row_mask:0x5 bank_mask:0x1" % (P.lo("inj"), cc), writes=[P.lo("inj")], sensitive=[cc])
        P.defer("v_lshl_add_u64 %s, %s, 0, %s" % (HI, HI, INJ), writes=[HI])
        P.emit("v_add_u32_dpp %s, %s, %s row_shl:1 %s" % (t, vl, vh, DPP_ODD), writes=[t], sensitive=[vl])
    else:
        wl = R("wl")
        P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (lo, hi, lo, DPP_ALL), writes=[lo], sensitive=[hi])      # w
        P.emit("v_lshrrev_b32 %s, %d, %s" % (vh, W, lo), writes=[vh])                                              # wh
        P.emit("v_and_b32 %s, %s, %s" % (wl, MASK, lo), writes=[wl])
        P.emit("v_add_u32 %s, %s, %s" % (cc, hi, vh), writes=[cc])
        P.emit("v_and_b32 %s, %s, %s" % (cc, ONLY15, cc), writes=[cc])       # this row's share of the carry into column NL
        P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (lo, vh, wl, DPP_ALL), writes=[lo], sensitive=[vh])       # w2
        P.emit("v_mov_b32 %s, %s" % (x, lo), writes=[x])
        P.emit("v_mov_b32_dpp %s, %s row_ror:1 row_mask:0xf bank_mask:0x1" % (P.lo("inj"), cc), writes=[P.lo("inj")], sensitive=[cc])
        P.emit("v_permlane16_swap_b32 %s, %s" % (lo, x), writes=[lo, x], sensitive=[lo, x])
        P.emit("v_add_u32 %s, %s, %s" % (lo, lo, x), writes=[lo])            # y
        P.emit("v_lshrrev_b32 %s, %d, %s" % (vh, W, lo), writes=[vh])        # yh
        P.emit("v_and_b32 %s, %s, %s" % (vl, MTOP, lo), writes=[vl])         # the top limb keeps its own carry
        P.defer("v_lshl_add_u64 %s, %s, 0, %s" % (HI, HI, INJ), writes=[HI])
        P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 row_mask:0x5 bank_mask:0xf bound_ctrl:1" % (t, vh, vl), writes=[t], sensitive=[vh])
        P.emit("v_add_u32_dpp %s, %s, %s row_shl:1 %s" % (t, vl, vh, DPP_ODD), writes=[t], sensitive=[vl])
    # ---- P2: fold ---------------------------------------------------------------------------------------------------
    tq = [R("a%d" % q) for q in range(Q)]            # the broadcast registers of P1 are free again
    for q in range(Q):
        P.emit("v_mov_b32_dpp %s, %s row_newbcast:%d %s" % (tq[q], t, OFF + 2 * q, DPP_ALL), writes=[tq[q]], sensitive=[t])
    P.flush()
    for q in range(Q):
        P.emit("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (HI, tq[q], CT[q], HI), writes=[HI])
    # ---- RN2 (fresh destination registers: the destination of a DPP instruction is hazard-sensitive too) -------------
    w, w2, yh, yl = R("w"), R("w2"), R("yh"), R("yl")
    if total_first:
        X2 = R("X2", pair=True)
        P.emit("v_mov_b32 %s, %s" % (P.lo("X2"), P.lo("HI")), writes=[P.lo("X2")])
        P.emit("v_mov_b32 %s, %s" % (P.hi("X2"), P.hi("HI")), writes=[P.hi("X2")])
        P.emit("v_permlane16_swap_b32 %s, %s" % (P.lo("HI"), P.lo("X2")), writes=[P.lo("HI"), P.lo("X2")],
               sensitive=[P.lo("HI"), P.lo("X2")])
        P.emit("v_permlane16_swap_b32 %s, %s" % (P.hi("HI"), P.hi("X2")), writes=[P.hi("HI"), P.hi("X2")],
               sensitive=[P.hi("HI"), P.hi("X2")])
        P.emit("v_lshl_add_u64 %s, %s, 0, %s" % (HI, HI, X2), writes=[HI])
        P.emit("v_alignbit_b32 %s, %s, %s, %d" % (hi, P.hi("HI"), P.lo("HI"), W), writes=[hi])
        P.emit("v_and_b32 %s, %s, %s" % (lo, MASK, P.lo("HI")), writes=[lo])
        P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (w, hi, lo, DPP_ALL), writes=[w], sensitive=[hi])
        P.emit("v_lshrrev_b32 %s, %d, %s" % (yh, W, w), writes=[yh])
        P.emit("v_and_b32 %s, %s, %s" % (yl, MASK, w), writes=[yl])
        P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (A, yh, yl, DPP_ALL), writes=[A], sensitive=[yh])
    else:
        P.emit("v_alignbit_b32 %s, %s, %s, %d" % (hi, P.hi("HI"), P.lo("HI"), W), writes=[hi])
        P.emit("v_and_b32 %s, %s, %s" % (lo, MASK, P.lo("HI")), writes=[lo])
        P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (w, hi, lo, DPP_ALL), writes=[w], sensitive=[hi])
        P.emit("v_lshrrev_b32 %s, %d, %s" % (vh, W, w), writes=[vh])
        P.emit("v_and_b32 %s, %s, %s" % (vl, MASK, w), writes=[vl])
        P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (w2, vh, vl, DPP_ALL), writes=[w2], sensitive=[vh])
        P.emit("v_mov_b32 %s, %s" % (x, w2), writes=[x])
        P.emit("v_permlane16_swap_b32 %s, %s" % (w2, x), writes=[w2, x], sensitive=[w2, x])
        P.emit("v_add_u32 %s, %s, %s" % (lo, w2, x), writes=[lo])            # y
        P.emit("v_lshrrev_b32 %s, %d, %s" % (yh, W, lo), writes=[yh])
        P.emit("v_and_b32 %s, %s, %s" % (yl, MASK, lo), writes=[yl])
        P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (A, yh, yl, DPP_ALL), writes=[A], sensitive=[yh])
    per_product = P.slot - body_start
    if loop:
        P.emit("s_cbranch_scc1 1b", salu=True)
        per_product += 1
    P.emit("s_nop 1", salu=True)      # the caller's next instruction may read the result through DPP
    clob = ["v%d" % i for i in range(VBASE, P.next_v)] + ["vcc"] + (["scc"] if loop else [])
    return P.lines, clob, dict(Q=Q, slots=per_product, nops=P.nops, cross_first=cross_first, total_first=total_first)


def render(nl, W):
    Q = (nl + 1) // 2
    out = []
    for kind, square, loop in (("mul", False, False), ("sqr_run", True, True)):
        lines, clob, info = gen_product(nl, W, square, loop)
        body = "\n".join('        "%s\\n\\t"' % l for l in lines)
        outs = ['"+v"(a)'] + (['"+s"(n)'] if loop else [])
        ins = ([] if square else ['"v"(b)']) + ['"v"(ct[%d])' % q for q in range(Q)]
        args = "uint32_t a, " + ("" if square else "uint32_t b, ") + "const uint32_t (&ct)[%d]" % Q
        if nl > 13:
            ins += ['"v"(mtop)', '"v"(only15)']
            args += ", uint32_t mtop, uint32_t only15"
        if loop:
            args += ", uint32_t n"
        out.append("  // %s: %d issue slots per product (%d of them s_nop)" % (kind, info["slots"], info["nops"]))
        out.append("  __device__ static __forceinline__ uint32_t %s(%s) {" % (kind, args))
        out.append("    asm volatile(\n%s\n        : %s\n        : %s\n        : %s);" % (
            body, ", ".join(outs), ", ".join(ins), ", ".join('"%s"' % c for c in clob)))
        out.append("    return a;\n  }")
    return out


def main():
    h = ["// GENERATED by tools/gen_coop2d_asm.py -- do not edit.  The two-row fold product of coop2d.h (tools/coop2d_model.py",
         "// is its specification) as hand-scheduled gfx950 assembly: a multiplication, and a run of squarings with the loop",
         "// inside the statement.  Layout-driven (limbs, limb bits): the fold table arrives as operands.",
         "#pragma once", "#include <hip/hip_runtime.h>", "#include <cstdint>", "namespace anemoi {",
         "template <int NL, int W> struct AsmCoop2d;"]
    for nl, W in LAYOUTS:
        h.append("template <> struct AsmCoop2d<%d, %d> {" % (nl, W))
        h += render(nl, W)
        h.append("};")
    h.append("}  // namespace anemoi")
    dst = os.path.join(ROOT, "anemoi-rust_amd", "csrc", "coop2d_asm_gen.h")
    with open(dst, "w") as f:
        f.write("\n".join(h) + "\n")
    print("wrote", dst)
    for nl, W in LAYOUTS:
        for square, loop in ((False, False), (True, True)):
            _, _, info = gen_product(nl, W, square, loop)
            print("  %2d limbs of %d bits, %s: %d slots per product, %d s_nop" % (nl, W, "squaring run" if square else "multiplication",
                                                                           info["slots"], info["nops"]))


if __name__ == "__main__":
    main()
