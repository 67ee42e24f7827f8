#!/usr/bin/env python3
"""Generate anemoi-rust_amd/csrc/coop2d_asm_gen.h: the two-row fold product of coop2d.h as hand-scheduled gfx950
assembly.  Three statements per layout:

    mul(a, b)              one multiplication
    sqr_run(a, n)          n >= 1 squarings, the loop inside the statement
    sqr_mul(a, n, b)       n >= 1 squarings and then a multiplication by b -- one step of the sliding-window
                           exponentiation; the operand-side work of the multiplication (the S form of b and its twelve
                           lane shifts, which do not depend on the squarings) is issued in the hazard gaps of the LAST
                           squaring

Why: a wavefront alone on its SIMD issues one instruction (or s_nop slot) per 4.08 cycles whatever the dependences
(tools/ubench/lone_wave_fetch.hip), so a product costs its issue slots.  hipcc's
version of Coop2d::mul is 94 slots per squaring on 11 limbs: 76 instructions + 18 hazard s_nop (it re-uses a handful of
registers, so nearly every DPP / permlane read waits for the write before it).  Scheduled by hand the hazard slots are
filled with work that is there anyway -- the HI multiply-adds of P1 are issued inside the carry chain of RN1, loop
control sits in front of the first DPP reads, the next multiplication's operand shifts in what is left.

The generator is layout-driven (NL, W): nothing depends on the modulus -- the fold table arrives as operands.
`Prog` keeps the issue order, knows which reads are hazard-sensitive (DPP sources and destinations, v_permlane16_swap
operands: a VGPR written by a VALU instruction may be read that way only two issue slots later) and fills a gap with
the next deferred instruction, or with s_nop where nothing is left; tests/test_coop2d_model.py executes the emitted
text on a 64-lane interpreter against the big-integer result, with overflow checks, and re-checks the hazard distances
independently.

    python tools/gen_coop2d_asm.py
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VBASE = 80          # clobbered VGPRs v[VBASE ...] (even: 64-bit pairs are 2-aligned)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asm_grid import align8, enc_size  # noqa: E402,F401

LAYOUTS = [(11, 27), (15, 28)]   # (limbs, limb bits) of the fold layouts in field_consts_gen.h
KINDS = ("mul", "sqr_run", "sqr_mul")
# A/B only: ANEMOI_COOP2D_GEN_NOVDST=1 writes csrc/coop2d_asm_gen_novdst.h without the padding of DPP destinations
PAD_DPP_DST = os.environ.get("ANEMOI_COOP2D_GEN_NOVDST") != "1"
# copies of the product in a run of squarings (one taken branch per run, not per squaring) and the alignment of the
# branch target behind it; the environment variables are for A/B builds only
UNROLL = int(os.environ.get("ANEMOI_COOP2D_GEN_UNROLL", "8"))
ALIGN_LOG2 = int(os.environ.get("ANEMOI_COOP2D_GEN_ALIGN", "5"))


class Prog:
    """Issue-ordered instruction list with automatic hazard padding and in-order gap filling."""

    def __init__(self):
        self.lines = []
        self.slot = 0
        self.wrote = {}          # vgpr name -> slot of its last VALU write
        self.nops = 0
        self.next_v = VBASE
        self.names = {}
        self.pending = []        # deferred independent instructions: issued, in order, where a hazard gap opens

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

    def _need(self, text, writes, sensitive):
        """issue slots still to pass before `text` may be issued"""
        if PAD_DPP_DST and "_dpp " in text:       # the destination of a DPP instruction is padded like a source (hipcc does: the lanes a
            sensitive = list(sensitive) + list(writes)   # row / bank mask disables keep their old value)
        need = 0
        for tok in sensitive:
            for r in self._vregs(tok):
                if r in self.wrote:
                    need = max(need, 2 - (self.slot - self.wrote[r] - 1))
        return need

    def _issue(self, text, writes, salu):
        self.lines.append(text)
        if not salu:
            for tok in writes:
                for r in self._vregs(tok):
                    self.wrote[r] = self.slot
        self.slot += 1

    def emit(self, text, writes=(), sensitive=(), salu=False):
        """`sensitive`: operands read through DPP or by a permlane swap."""
        need = self._need(text, writes, sensitive)
        while need > 0 and self.pending and self._need(*self.pending[0][:3]) == 0:
            f = self.pending.pop(0)          # a deferred instruction is a better filler than s_nop
            self._issue(f[0], f[1], False)
            need = self._need(text, writes, sensitive)
        if need > 0:
            self.lines.append("s_nop %d" % (need - 1))
            self.slot += need
            self.nops += need
        self._issue(text, writes, salu)

    def defer(self, text, writes=(), sensitive=(), tag="hi"):
        self.pending.append((text, list(writes), list(sensitive), tag))

    def flush(self, tag=None):
        todo = [f for f in self.pending if tag is None or f[3] == tag]
        keep = [f for f in self.pending if not (tag is None or f[3] == tag)]
        self.pending = []                    # (emit() must not pull from the queue while it is being drained)
        for f in todo:
            self.emit(f[0], writes=f[1], sensitive=f[2])
        self.pending = keep

    def barrier(self):
        for r in list(self.wrote):           # every predecessor: assume everything was written in the previous slot
            self.wrote[r] = self.slot - 1

    def label(self, name):
        self.lines.append(name + ":")
        self.barrier()


DPP_ALL = "row_mask:0xf bank_mask:0xf bound_ctrl:1"
DPP_ODD = "row_mask:0xa bank_mask:0xf bound_ctrl:1"
DPP_EVEN = "row_mask:0x5 bank_mask:0xf bound_ctrl:1"


def operand_names(nl, kind, rows=2):
    """GCC operand numbers of a statement: outputs first (%0 = a in/out, then the SGPR run length of the statements
    that square), then the inputs: [b], the Q fold-table registers, [for 15 limbs: the limb mask with lane 15 all ones,
    and the mask that is all ones in lane 15 only]."""
    Q = (nl + rows - 1) // rows
    names, k = {"A": "%0"}, 1
    if kind != "mul":
        names["CNT"], k = "%1", 2
    if kind != "sqr_run":
        names["B"], k = "%%%d" % k, k + 1
    names["CT"] = ["%%%d" % (k + q) for q in range(Q)]
    k += Q
    if nl > 13:
        names["MTOP"], names["ONLY15"] = "%%%d" % k, "%%%d" % (k + 1)
    return names


class Gen:
    def __init__(self, nl, W, kind):
        self.nl, self.W, self.kind = nl, W, kind
        self.Q, self.OFF = (nl + 1) // 2, 16 - nl
        self.cross_first = nl <= 13      # RN1: sum over the rows first (the sum of the high parts <= NL 2^W fits 32 bits)
        self.total_first = W <= 27       # RN2: 64-bit sum over the rows first (21 products of 2^54 leave a 32-bit carry)
        self.P = Prog()
        self.ops = operand_names(nl, kind)
        self.MASK = "0x%x" % ((1 << W) - 1)   # a literal costs no issue slot
        P = self.P
        self.INJ = P.reg("inj", pair=True)   # [carry | 0]: only lanes 0..3 are ever rewritten, the rest stays 0
        self.LO, self.HI = P.reg("LO", pair=True), P.reg("HI", pair=True)

    # one product: out = a * b R'^-1.  `a`: register with the multiplier (plain form).  `b`: register with the multiplicand
    # (plain form; None = squaring, then `a` is rewritten in place), or `shifts` = (bl[q], bh[q]) prepared beforehand.
    # `ctl`: SALU instructions (loop control) to issue in the head's hazard gaps.  `after_hi`: called once the HI
    # multiply-adds are queued, to queue further independent work behind them.
    def product(self, a, b, out, shifts=None, ctl=(), after_hi=None):
        P, R, nl, W, Q, OFF, MASK = self.P, self.P.reg, self.nl, self.W, self.Q, self.OFF, self.MASK
        LO, HI, INJ = self.LO, self.HI, self.INJ
        ctl = list(ctl)
        AD = R("aD")
        P.emit("v_mov_b32 %s, %s" % (AD, a), writes=[AD])
        while ctl:                           # (together: a pair of 4-byte SALU instructions stays on the 8-byte fetch grid)
            P.emit(ctl.pop(0), salu=True)
        BS = None
        if shifts is None:
            BS = a if b is None else R("bS")
            if b is not None:
                P.emit("v_mov_b32 %s, %s" % (BS, b), writes=[BS])
        P.emit("v_mov_b32_dpp %s, %s row_shl:1 %s" % (AD, a, DPP_ODD), writes=[AD], sensitive=[a])
        if shifts is None:
            P.emit("v_mov_b32_dpp %s, %s row_shr:1 %s" % (BS, b or a, DPP_ODD), writes=[BS], sensitive=[b or a])
        while ctl:
            P.emit(ctl.pop(0), salu=True)
        if shifts is not None:
            P.flush("pre")                   # whatever is left of the prefetched operand shifts
        for q in range(Q):
            aq = R("a%d" % q)
            P.emit("v_mov_b32_dpp %s, %s row_newbcast:%d %s" % (aq, AD, 2 * q, DPP_ALL), writes=[aq], sensitive=[AD])
            if shifts is None:
                bl, bh = R("bl%d" % (q & 1)), R("bh%d" % q)
                P.emit("v_mov_b32_dpp %s, %s row_shr:%d %s" % (bl, BS, OFF + 2 * q, DPP_ALL), writes=[bl], sensitive=[BS])
                assert 1 <= nl - 2 * q <= 15
                P.emit("v_mov_b32_dpp %s, %s row_shl:%d %s" % (bh, BS, nl - 2 * q, DPP_ALL), writes=[bh], sensitive=[BS])
            else:
                bl, bh = shifts[0][q], shifts[1][q]
            P.emit("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (LO, aq, bl, LO if q else "0"), writes=[LO])
            # the HI multiply-adds are deferred: they fill the hazard gaps of RN1's carry chain
            P.defer("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (HI, aq, bh, HI if q else "0"), writes=[HI], tag="hi")
        if after_hi:
            after_hi()
        # ---- RN1: low columns -> limbs t (D form); the carry out of column NL-1 -> HI lane 0 ------------------------
        lo, hi, x = R("lo"), R("hi"), R("x")
        vh, vl, cc, t = R("vh"), R("vl"), R("cc"), R("t")
        P.emit("v_alignbit_b32 %s, %s, %s, %d" % (hi, P.hi("LO"), P.lo("LO"), W), writes=[hi])
        P.emit("v_and_b32 %s, %s, %s" % (lo, MASK, P.lo("LO")), writes=[lo])
        if self.cross_first:
            P.emit("v_permlane16_swap_b32 %s, %s" % (lo, hi), writes=[lo, hi], sensitive=[lo, hi])
            P.emit("v_add_u32 %s, %s, %s" % (lo, lo, hi), writes=[lo])       # even row: sum of lo; odd row: sum of hi
            P.emit("v_mov_b32 %s, %s" % (x, lo), writes=[x])
            P.emit("v_permlane16_swap_b32 %s, %s" % (lo, x), writes=[lo, x], sensitive=[lo, x])   # lo = low parts, x = high parts
            P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (hi, x, lo, DPP_ALL), writes=[hi], sensitive=[x])   # v
            P.emit("v_lshrrev_b32 %s, %d, %s" % (vh, W, hi), writes=[vh])
            P.emit("v_add_u32 %s, %s, %s" % (cc, x, vh), writes=[cc])         # lane 15: all that column NL-1 hands on
            P.emit("v_and_b32 %s, %s, %s" % (vl, MASK, hi), writes=[vl])
            # t in D form: even rows t_l = vl_l + vh_(l-1); odd rows t_(l+1) = vl_(l+1) + vh_l
            P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (t, vh, vl, DPP_EVEN), writes=[t], sensitive=[vh])
            P.emit("v_mov_b32_dpp %s, %s row_ror:1 row_mask:0x5 bank_mask:0x1" % (P.lo("inj"), cc), writes=[P.lo("inj")], sensitive=[cc])
            P.defer("v_lshl_add_u64 %s, %s, 0, %s" % (HI, HI, INJ), writes=[HI], tag="hi")
            P.emit("v_add_u32_dpp %s, %s, %s row_shl:1 %s" % (t, vl, vh, DPP_ODD), writes=[t], sensitive=[vl])
        else:
            wl = R("wl")
            MTOP, ONLY15 = self.ops["MTOP"], self.ops["ONLY15"]
            P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (lo, hi, lo, DPP_ALL), writes=[lo], sensitive=[hi])      # w
            P.emit("v_lshrrev_b32 %s, %d, %s" % (vh, W, lo), writes=[vh])                                              # wh
            P.emit("v_and_b32 %s, %s, %s" % (wl, MASK, lo), writes=[wl])
            P.emit("v_add_u32 %s, %s, %s" % (cc, hi, vh), writes=[cc])
            P.emit("v_and_b32 %s, %s, %s" % (cc, ONLY15, cc), writes=[cc])   # this row's share of the carry into column NL
            P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (lo, vh, wl, DPP_ALL), writes=[lo], sensitive=[vh])       # w2
            P.emit("v_mov_b32 %s, %s" % (x, lo), writes=[x])
            P.emit("v_mov_b32_dpp %s, %s row_ror:1 row_mask:0xf bank_mask:0x1" % (P.lo("inj"), cc), writes=[P.lo("inj")], sensitive=[cc])
            P.emit("v_permlane16_swap_b32 %s, %s" % (lo, x), writes=[lo, x], sensitive=[lo, x])
            P.emit("v_add_u32 %s, %s, %s" % (lo, lo, x), writes=[lo])        # y
            P.emit("v_lshrrev_b32 %s, %d, %s" % (vh, W, lo), writes=[vh])    # yh
            P.emit("v_and_b32 %s, %s, %s" % (vl, MTOP, lo), writes=[vl])     # the top limb keeps its own carry
            P.defer("v_lshl_add_u64 %s, %s, 0, %s" % (HI, HI, INJ), writes=[HI], tag="hi")
            P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (t, vh, vl, DPP_EVEN), writes=[t], sensitive=[vh])
            P.emit("v_add_u32_dpp %s, %s, %s row_shl:1 %s" % (t, vl, vh, DPP_ODD), writes=[t], sensitive=[vl])
        # ---- P2: fold -----------------------------------------------------------------------------------------------
        tq = [R("a%d" % q) for q in range(Q)]        # the broadcast registers of P1 are free again ...
        P.flush("hi")                                # ... once the last HI multiply-add has read them
        for q in range(Q):
            P.emit("v_mov_b32_dpp %s, %s row_newbcast:%d %s" % (tq[q], t, OFF + 2 * q, DPP_ALL), writes=[tq[q]], sensitive=[t])
        for q in range(Q):
            P.emit("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (HI, tq[q], self.ops["CT"][q], HI), writes=[HI])
        # ---- RN2 (fresh destination registers: the destination of a DPP instruction is hazard-sensitive too) ---------
        w, w2, yh, yl = R("w"), R("w2"), R("yh"), R("yl")
        if self.total_first:
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
            P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (out, yh, yl, DPP_ALL), writes=[out], sensitive=[yh])
        else:
            P.emit("v_alignbit_b32 %s, %s, %s, %d" % (hi, P.hi("HI"), P.lo("HI"), W), writes=[hi])
            P.emit("v_and_b32 %s, %s, %s" % (lo, MASK, P.lo("HI")), writes=[lo])
            P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (w, hi, lo, DPP_ALL), writes=[w], sensitive=[hi])
            P.emit("v_lshrrev_b32 %s, %d, %s" % (vh, W, w), writes=[vh])
            P.emit("v_and_b32 %s, %s, %s" % (vl, MASK, w), writes=[vl])
            P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (w2, vh, vl, DPP_ALL), writes=[w2], sensitive=[vh])
            P.emit("v_mov_b32 %s, %s" % (x, w2), writes=[x])
            P.emit("v_permlane16_swap_b32 %s, %s" % (w2, x), writes=[w2, x], sensitive=[w2, x])
            P.emit("v_add_u32 %s, %s, %s" % (lo, w2, x), writes=[lo])        # y
            P.emit("v_lshrrev_b32 %s, %d, %s" % (yh, W, lo), writes=[yh])
            P.emit("v_and_b32 %s, %s, %s" % (yl, MASK, lo), writes=[yl])
            P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (out, yh, yl, DPP_ALL), writes=[out], sensitive=[yh])

    def queue_operand_shifts(self, b):
        """the S form of b and its 2 Q lane shifts, queued as gap fillers (tag "pre"); returns (bl[q], bh[q])"""
        P, R, nl, Q, OFF = self.P, self.P.reg, self.nl, self.Q, self.OFF
        mbs = R("mbS")
        P.defer("v_mov_b32 %s, %s" % (mbs, b), writes=[mbs], tag="pre")
        P.defer("v_mov_b32_dpp %s, %s row_shr:1 %s" % (mbs, b, DPP_ODD), writes=[mbs], sensitive=[b], tag="pre")
        bl, bh = [R("mbl%d" % q) for q in range(Q)], [R("mbh%d" % q) for q in range(Q)]
        for q in range(Q):
            P.defer("v_mov_b32_dpp %s, %s row_shr:%d %s" % (bl[q], mbs, OFF + 2 * q, DPP_ALL), writes=[bl[q]], sensitive=[mbs], tag="pre")
            P.defer("v_mov_b32_dpp %s, %s row_shl:%d %s" % (bh[q], mbs, nl - 2 * q, DPP_ALL), writes=[bh[q]], sensitive=[mbs], tag="pre")
        return bl, bh

    def squarings(self, A, CNT, done, info):
        """CNT (>= 1) squarings of A, then on to label `done` (which follows).  A taken branch costs a lone wavefront an
        instruction refetch -- tens of cycles, more when the target sits at the end of a fetch line (measured: aligning
        the loop head alone was worth 2-7 %) -- so the run is UNROLL copies of the product, each followed by an exit
        branch that is NOT taken until the count is used up: one taken branch per run instead of one per squaring.
        Runs longer than UNROLL go round; `done` is aligned to a fetch line (the padding is jumped over)."""
        P = self.P
        P.label("1")
        for u in range(UNROLL):
            if u:
                P.barrier()
            s0 = P.slot
            self.product(A, None, A, ctl=["s_sub_u32 %s, %s, 1" % (CNT, CNT), "s_cmp_eq_u32 %s, 0" % CNT])
            P.emit("s_cbranch_scc1 %sf" % done, salu=True)
            if not u:
                info["slots"] = P.slot - s0
        P.emit("s_branch 1b", salu=True)
        P.lines.append(".p2align %d" % ALIGN_LOG2)

    def build(self):
        P, ops, kind = self.P, self.ops, self.kind
        A, B, CNT = ops["A"], ops.get("B"), ops.get("CNT")
        P.emit("v_mov_b32 %s, 0" % P.lo("inj"), writes=[P.lo("inj")])
        P.emit("v_mov_b32 %s, 0" % P.hi("inj"), writes=[P.hi("inj")])
        P.wrote[A] = P.slot - 1           # whatever wrote the operands last: assume the slot before the statement
        if B:
            P.wrote[B] = P.slot - 1
        info = {}
        if kind == "mul":
            s0 = P.slot
            self.product(A, B, A)
            info["slots"] = P.slot - s0
        elif kind == "sqr_run":
            self.squarings(A, CNT, "9", info)
            P.label("9")
        else:  # sqr_mul: n - 1 squarings, then the last one, which carries the multiplication's operand shifts
            P.emit("s_sub_u32 %s, %s, 1" % (CNT, CNT), salu=True)
            P.emit("s_cmp_eq_u32 %s, 0" % CNT, salu=True)
            P.emit("s_cbranch_scc1 2f", salu=True)
            self.squarings(A, CNT, "2", info)
            P.label("2")
            s1 = P.slot
            shifts = {}
            self.product(A, None, A, after_hi=lambda: shifts.update(v=self.queue_operand_shifts(B)))
            self.product(A, None, A, shifts=shifts["v"])
            info["last_sqr_plus_mul_slots"] = P.slot - s1
        P.emit("s_nop 1", salu=True)      # the caller's next instruction may read the result through DPP
        clob = ["v%d" % i for i in range(VBASE, P.next_v)] + ["vcc"] + (["scc"] if kind != "mul" else [])
        info.update(Q=self.Q, nops=P.nops, cross_first=self.cross_first, total_first=self.total_first)
        return P.lines, clob, info


class Gen4(Gen):
    """Four rows per element, ONE element per wavefront (11-limb fields: the S form shifts b up by up to three lanes, so
    NL + 3 <= 16).  Row r multiplies by the limbs a_i with i = r (mod 4): Q4 = ceil(NL / 4) steps per phase -- half the
    multiply-adds, broadcasts and shifts of the two-row form per wavefront, paid for with a second swap level
    (v_permlane32_swap) in the two sums over the rows.  tools/coop2d_model.py::mul4 is the specification."""

    def __init__(self, nl, W, kind):
        super().__init__(nl, W, kind)
        assert nl <= 13 and W <= 27
        self.Q = (nl + 3) // 4
        self.ops = operand_names(nl, kind, rows=4)

    def product(self, a, b, out, shifts=None, ctl=(), after_hi=None):
        P, R, nl, W, Q, OFF, MASK = self.P, self.P.reg, self.nl, self.W, self.Q, self.OFF, self.MASK
        LO, HI, INJ = self.LO, self.HI, self.INJ
        ctl = list(ctl)
        row = lambda r: "row_mask:0x%x bank_mask:0xf bound_ctrl:1" % (1 << r)
        AD = R("aD")
        P.emit("v_mov_b32 %s, %s" % (AD, a), writes=[AD])
        while ctl:
            P.emit(ctl.pop(0), salu=True)
        BS = None
        if shifts is None:
            BS = R("bS")          # a copy also when squaring: rewriting `a` in place would put a VALU write of `a` in
            P.emit("v_mov_b32 %s, %s" % (BS, b or a), writes=[BS])   # front of every D-form read of it
        # D form: row r lane l holds a_(l+r); S form: row r lane l holds b_(l-r).  The two registers are written
        # alternately (a DPP destination is hazard-padded like a source), loop control in the gaps.
        for r in (1, 2, 3):
            P.emit("v_mov_b32_dpp %s, %s row_shl:%d %s" % (AD, a, r, row(r)), writes=[AD], sensitive=[a])
            if shifts is None:
                P.emit("v_mov_b32_dpp %s, %s row_shr:%d %s" % (BS, b or a, r, row(r)), writes=[BS], sensitive=[b or a])
            if ctl:
                P.emit(ctl.pop(0), salu=True)
        while ctl:
            P.emit(ctl.pop(0), salu=True)
        if shifts is not None:
            P.flush("pre")
        for q in range(Q):
            aq = R("a%d" % q)
            P.emit("v_mov_b32_dpp %s, %s row_newbcast:%d %s" % (aq, AD, 4 * q, DPP_ALL), writes=[aq], sensitive=[AD])
            if shifts is None:
                bl, bh = R("bl%d" % (q & 1)), R("bh%d" % q)
                P.emit("v_mov_b32_dpp %s, %s row_shr:%d %s" % (bl, BS, OFF + 4 * q, DPP_ALL), writes=[bl], sensitive=[BS])
                assert 1 <= nl - 4 * q <= 15 and OFF + 4 * q <= 15
                P.emit("v_mov_b32_dpp %s, %s row_shl:%d %s" % (bh, BS, nl - 4 * q, DPP_ALL), writes=[bh], sensitive=[BS])
            else:
                bl, bh = shifts[0][q], shifts[1][q]
            P.emit("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (LO, aq, bl, LO if q else "0"), writes=[LO])
            P.defer("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (HI, aq, bh, HI if q else "0"), writes=[HI], tag="hi")
        if after_hi:
            after_hi()
        # ---- RN1: the low columns summed over the four rows (two swap levels), carried to limbs t in D form ----------
        lo, hi, x, y = R("lo"), R("hi"), R("x"), R("y4")
        vh, vl, cc, t = R("vh"), R("vl"), R("cc"), R("t")
        P.emit("v_alignbit_b32 %s, %s, %s, %d" % (hi, P.hi("LO"), P.lo("LO"), W), writes=[hi])
        P.emit("v_and_b32 %s, %s, %s" % (lo, MASK, P.lo("LO")), writes=[lo])
        P.emit("v_permlane16_swap_b32 %s, %s" % (lo, hi), writes=[lo, hi], sensitive=[lo, hi])
        P.emit("v_add_u32 %s, %s, %s" % (lo, lo, hi), writes=[lo])           # rows: lo0+lo1, hi0+hi1, lo2+lo3, hi2+hi3
        P.emit("v_mov_b32 %s, %s" % (x, lo), writes=[x])
        P.emit("v_permlane32_swap_b32 %s, %s" % (lo, x), writes=[lo, x], sensitive=[lo, x])
        P.emit("v_add_u32 %s, %s, %s" % (lo, lo, x), writes=[lo])            # rows: sum lo, sum hi, sum lo, sum hi
        P.emit("v_mov_b32 %s, %s" % (y, lo), writes=[y])
        P.emit("v_permlane16_swap_b32 %s, %s" % (lo, y), writes=[lo, y], sensitive=[lo, y])   # lo = low parts, y = high parts
        P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (hi, y, lo, DPP_ALL), writes=[hi], sensitive=[y])   # v
        P.emit("v_lshrrev_b32 %s, %d, %s" % (vh, W, hi), writes=[vh])
        P.emit("v_add_u32 %s, %s, %s" % (cc, y, vh), writes=[cc])
        P.emit("v_and_b32 %s, %s, %s" % (vl, MASK, hi), writes=[vl])
        P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (t, vh, vl, DPP_ALL), writes=[t], sensitive=[vh])
        P.emit("v_mov_b32_dpp %s, %s row_ror:1 row_mask:0x1 bank_mask:0x1" % (P.lo("inj"), cc), writes=[P.lo("inj")], sensitive=[cc])
        P.defer("v_lshl_add_u64 %s, %s, 0, %s" % (HI, HI, INJ), writes=[HI], tag="hi")
        for r in (1, 2, 3):       # t in D form, in place
            P.emit("v_mov_b32_dpp %s, %s row_shl:%d %s" % (t, t, r, row(r)), writes=[t], sensitive=[t])
        # ---- P2 -------------------------------------------------------------------------------------------------------
        tq = [R("a%d" % q) for q in range(Q)]
        P.flush("hi")
        for q in range(Q):
            P.emit("v_mov_b32_dpp %s, %s row_newbcast:%d %s" % (tq[q], t, OFF + 4 * q, DPP_ALL), writes=[tq[q]], sensitive=[t])
        for q in range(Q):
            P.emit("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (HI, tq[q], self.ops["CT"][q], HI), writes=[HI])
        # ---- RN2: the 64-bit sum over the four rows, two carry passes --------------------------------------------------
        w, yh, yl = R("w"), R("yh"), R("yl")
        X2 = R("X2", pair=True)
        for swap in ("v_permlane16_swap_b32", "v_permlane32_swap_b32"):
            P.emit("v_mov_b32 %s, %s" % (P.lo("X2"), P.lo("HI")), writes=[P.lo("X2")])
            P.emit("v_mov_b32 %s, %s" % (P.hi("X2"), P.hi("HI")), writes=[P.hi("X2")])
            P.emit("%s %s, %s" % (swap, P.lo("HI"), P.lo("X2")), writes=[P.lo("HI"), P.lo("X2")], sensitive=[P.lo("HI"), P.lo("X2")])
            P.emit("%s %s, %s" % (swap, P.hi("HI"), P.hi("X2")), writes=[P.hi("HI"), P.hi("X2")], sensitive=[P.hi("HI"), P.hi("X2")])
            P.emit("v_lshl_add_u64 %s, %s, 0, %s" % (HI, HI, X2), writes=[HI])
        P.emit("v_alignbit_b32 %s, %s, %s, %d" % (hi, P.hi("HI"), P.lo("HI"), W), writes=[hi])
        P.emit("v_and_b32 %s, %s, %s" % (lo, MASK, P.lo("HI")), writes=[lo])
        P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (w, hi, lo, DPP_ALL), writes=[w], sensitive=[hi])
        P.emit("v_lshrrev_b32 %s, %d, %s" % (yh, W, w), writes=[yh])
        P.emit("v_and_b32 %s, %s, %s" % (yl, MASK, w), writes=[yl])
        P.emit("v_add_u32_dpp %s, %s, %s row_shr:1 %s" % (out, yh, yl, DPP_ALL), writes=[out], sensitive=[yh])

    def queue_operand_shifts(self, b):
        P, R, nl, Q, OFF = self.P, self.P.reg, self.nl, self.Q, self.OFF
        mbs = R("mbS")
        P.defer("v_mov_b32 %s, %s" % (mbs, b), writes=[mbs], tag="pre")
        for r in (1, 2, 3):
            P.defer("v_mov_b32_dpp %s, %s row_shr:%d row_mask:0x%x bank_mask:0xf bound_ctrl:1" % (mbs, b, r, 1 << r),
                    writes=[mbs], sensitive=[b], tag="pre")
        bl, bh = [R("mbl%d" % q) for q in range(Q)], [R("mbh%d" % q) for q in range(Q)]
        for q in range(Q):
            P.defer("v_mov_b32_dpp %s, %s row_shr:%d %s" % (bl[q], mbs, OFF + 4 * q, DPP_ALL), writes=[bl[q]], sensitive=[mbs], tag="pre")
            P.defer("v_mov_b32_dpp %s, %s row_shl:%d %s" % (bh[q], mbs, nl - 4 * q, DPP_ALL), writes=[bh[q]], sensitive=[mbs], tag="pre")
        return bl, bh


# ---- fetch alignment: tools/asm_grid.py (every 8-byte instruction of a statement starts on an 8-byte boundary) ---------
ALIGN8 = os.environ.get("ANEMOI_COOP2D_GEN_ALIGN8", "1") == "1"    # (0: A/B builds only)


def gen_product(nl, W, kind, rows=2):
    """-> (lines, clobbers, info) of the statement `kind` in KINDS"""
    lines, clob, info = (Gen4 if rows == 4 else Gen)(nl, W, kind).build()
    if ALIGN8:
        lines, info["off_grid"] = align8(lines, vnop=True)   # (<= one wavefront per SIMD: v_nop_e64 may stand in for s_nop 0)
    return lines, clob, info


def inputs_read_after_first_write(lines, nl, kind, rows=2):
    """the input operands (b, the fold table, the masks) that an instruction still READS after the first instruction that
    WRITES %0 -- if there is any, %0 must be an early-clobber operand"""
    names = operand_names(nl, kind, rows)
    inputs = [names[k] for k in ("B", "MTOP", "ONLY15") if k in names] + list(names["CT"])
    written, late = False, set()
    for ln in lines:
        if ln.startswith(".") or ln.endswith(":"):
            continue
        op, _, rest = ln.partition(" ")
        args = [x.strip() for x in re.split(r",\s*(?![^\[]*\])", rest)] if rest else []
        toks = [re.split(r"\s", a)[0] for a in args]
        if written:
            late.update(t for t in toks[1:] if t in inputs)
            # (a branch back into the run re-reads everything the loop body reads: bodies follow their first write)
        if toks and toks[0] == names["A"] and not op.startswith(("s_", "v_cmp", "v_nop")):
            written = True
    return sorted(late)


def render(nl, W, rows=2):
    Q = (nl + rows - 1) // rows
    out = []
    for kind in KINDS:
        lines, clob, info = gen_product(nl, W, kind, rows)
        body = "\n".join('        "%s\\n\\t"' % l for l in lines)
        # `a` is EARLY-CLOBBER: every statement writes %0 long before it has read its last input (sqr_mul reads b in its
        # last squaring, after the whole run has rewritten a; all of them read the fold table to the end).  Declared as a
        # plain "+v", nothing stops the register allocator from giving `a` and an input that holds the SAME VALUE one
        # register -- the leading-run doubling does `acc = sqr_mul(acc, n, tmp)` right after `tmp = acc` -- and the
        # result would be silently wrong (round-4 advisor finding; inputs_read_after_first_write() derives the need
        # from the emitted text, tests/test_coop2d_model.py pins it).
        early = inputs_read_after_first_write(lines, nl, kind, rows)
        outs = ['"+%sv"(a)' % ("&" if early else "")] + (['"+s"(n)'] if kind != "mul" else [])
        ins = (['"v"(b)'] if kind != "sqr_run" else []) + ['"v"(ct[%d])' % q for q in range(Q)]
        args = "uint32_t a, " + ("uint32_t b, " if kind != "sqr_run" else "") + "const uint32_t (&ct)[%d]" % Q
        if nl > 13:
            ins += ['"v"(mtop)', '"v"(only15)']
            args += ", uint32_t mtop, uint32_t only15"
        if kind != "mul":
            args += ", uint32_t n"
        what = ("%d issue slots" % info["slots"]) if kind == "mul" else ("%d issue slots per squaring" % info["slots"])
        if kind == "sqr_mul":
            what += ", %d for the last squaring + the multiplication" % info["last_sqr_plus_mul_slots"]
        out.append("  // %s: %s (%d s_nop in the whole statement)" % (kind, what, info["nops"]))
        out.append("  __device__ static __forceinline__ uint32_t %s(%s) {" % (kind, args))
        out.append("    asm volatile(\n%s\n        : %s\n        : %s\n        : %s);" % (
            body, ", ".join(outs), ", ".join(ins), ", ".join('"%s"' % c for c in clob)))
        out.append("    return a;\n  }")
    return out


def main():
    h = ["// GENERATED by tools/gen_coop2d_asm.py -- do not edit.  The two-row fold product of coop2d.h (tools/coop2d_model.py",
         "// is its specification) as hand-scheduled gfx950 assembly: a multiplication, a run of squarings, and a run of",
         "// squarings followed by a multiplication (one exponentiation step), the loops inside the statements.",
         "// Layout-driven (limbs, limb bits): the fold table arrives as operands.",
         "// The four-row statements (ROWS = 4, the recorded negative) are compiled into `make AB=1` libraries only.",
         "#pragma once", "#include <hip/hip_runtime.h>", "#include <cstdint>", '#include "build_config.h"', "namespace anemoi {",
         "template <int NL, int W, int ROWS = 2> struct AsmCoop2d;   // ROWS 16-lane rows per element"]
    for nl, W in LAYOUTS:
        for rows in ((2, 4) if nl <= 13 else (2,)):
            if rows == 4:
                h.append("#if ANEMOI_AB_BUILD")
            h.append("template <> struct AsmCoop2d<%d, %d, %d> {" % (nl, W, rows))
            h += render(nl, W, rows)
            h.append("};")
            if rows == 4:
                h.append("#endif  // ANEMOI_AB_BUILD")
    h.append("}  // namespace anemoi")
    dst = os.path.join(ROOT, "anemoi-rust_amd", "csrc", os.environ.get("ANEMOI_COOP2D_GEN_OUT") or
                       ("coop2d_asm_gen.h" if PAD_DPP_DST else "coop2d_asm_gen_novdst.h"))
    with open(dst, "w") as f:
        f.write("\n".join(h) + "\n")
    print("wrote", dst)
    for nl, W in LAYOUTS:
        for rows in ((2, 4) if nl <= 13 else (2,)):
            for kind in KINDS:
                _, _, info = gen_product(nl, W, kind, rows)
                print("  %2d limbs of %d bits, %d rows, %-8s %s" % (nl, W, rows, kind, {k: v for k, v in info.items() if k in ("slots", "last_sqr_plus_mul_slots", "nops", "off_grid")}))


if __name__ == "__main__":
    main()
