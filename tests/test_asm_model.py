"""Executes the GENERATED gfx950 assembly of the Montgomery squaring / multiplication
(anemoi-rust_amd/csrc/mont29_asm_gen.h, tools/gen_asm_mul.py) on a small CPU interpreter of the few
instructions it uses, with overflow detection on every v_mad_u64_u32 / v_lshl_add_u64 and on the 32-bit
column-carry path (v_alignbit_b32: the carry must really fit 32 bits).

Why: the GPU parity tests feed random field elements, which never come near the worst-case limb
patterns the 64-bit column accumulators are dimensioned for (30-bit limbs: a column holds up to 26
products of 2^60 and is split where it could overflow).  Here every statement is run on adversarial
inputs -- all limbs at their maximum, the top limb at its documented bound -- and the result must be
the Montgomery product (value-level, any representative below 2^(W NL)) with limbs below 2^W.
CPU only, no GPU and no oracle involved: this pins the generator's bound bookkeeping.
"""
import json
import os
import random
import re
import sys

import pytest

from conftest import FIELD_IDS, ROOT

HDR = os.path.join(ROOT, "anemoi-rust_amd", "csrc", "mont29_asm_gen.h")
M32, M64 = (1 << 32) - 1, (1 << 64) - 1


def parse_header():
    """{(field_id, W): {"NL": nl, "sqr": (lines, outs, ins), "mul": (lines, outs, ins)}} from the generated header: the
    bodies are string macros (ANEMOI_ASM_{SQR,MUL}_BODY_f_W) shared by the struct's functions and the ANEMOI_PIN_* macros;
    the squaring's inputs are the limbs of p and n0inv (ANEMOI_ASM_SQR_INS_f_W, SGPR operands), the multiplication's the
    limbs of b (VGPR operands)"""
    text = open(HDR).read()
    out = {}
    for m in re.finditer(r"template <> struct AsmMont<(\d+), (\d+)> \{(.*?)\n\};", text, re.S):
        fid, W, body = int(m.group(1)), int(m.group(2)), m.group(3)
        nl = int(re.search(r"NL = (\d+);", body).group(1))
        entry = {"NL": nl}
        outs = [("a", str(i)) for i in range(nl)]
        for name in ("sqr", "mul"):
            mb = re.search(r"#define ANEMOI_ASM_%s_BODY_%d_%d \\\n((?:[ \t]+\"[^\n]*\n)+)" % (name.upper(), fid, W), text)
            lines = re.findall(r'"([^"]+?)\\n\\t"', mb.group(1))
            if name == "sqr":
                mi = re.search(r"#define ANEMOI_ASM_SQR_INS_%d_%d (.*)" % (fid, W), text).group(1)
                ins = [("s", h, "", "") for h in re.findall(r'"s"\(0x([0-9a-f]+)u\)', mi)]
            else:
                ins = [("v", "", "b", str(i)) for i in range(nl)]
            entry[name] = (lines, outs, ins)
        out[(fid, W)] = entry
    return out


class Overflow(Exception):
    pass


def run(lines, outs, ins, a, b=None):
    """interpret one asm statement; a, b = limb lists; returns the new a"""
    ops = {}  # "%k" -> ("a", i) | ("b", i) | ("const", value)
    for k, (_, i) in enumerate(outs):
        ops["%%%d" % k] = ("a", int(i))
    for k, (_, hexv, arr, idx) in enumerate(ins):
        ops["%%%d" % (len(outs) + k)] = ("const", int(hexv, 16)) if hexv else ("b", int(idx))
    a, b = list(a), list(b) if b is not None else None
    reg = {}

    def rd(tok):
        tok = tok.strip()
        if tok in ops:
            kind, v = ops[tok]
            return a[v] if kind == "a" else b[v] if kind == "b" else v
        if tok.startswith("0x"):
            return int(tok, 16)
        if re.fullmatch(r"-?\d+", tok):
            return int(tok) & M32
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            lo = int(m.group(1))
            return reg.get("v%d" % lo, 0) | (reg.get("v%d" % (lo + 1), 0) << 32)
        return reg[tok]  # vN / sN: KeyError = read before write

    def wr(tok, val):
        tok = tok.strip()
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            lo = int(m.group(1))
            reg["v%d" % lo], reg["v%d" % (lo + 1)] = val & M32, (val >> 32) & M32
        elif tok in ops:
            kind, v = ops[tok]
            assert kind == "a"
            a[v] = val & M32
        else:
            reg[tok] = val & M32

    for ln in lines:
        if ln.startswith(".p2align") or ln.startswith("v_nop") or ln.startswith("s_nop"):
            continue               # layout on the 8-byte fetch grid (tools/asm_grid.py), not semantics
        op, rest = ln.split(None, 1)
        op = re.sub(r"_e(32|64)$", "", op)
        # split operands, keeping v[lo:hi] together
        args = [x.strip() for x in re.split(r",\s*(?![^\[]*\])", rest)]
        if op == "v_mad_u64_u32":
            d, _vcc, s0, s1, s2 = args
            val = rd(s0) * rd(s1) + rd(s2)
            if val > M64:
                raise Overflow(ln)
            wr(d, val)
        elif op == "v_lshl_add_u64":
            d, s0, sh, s2 = args
            val = (rd(s0) << rd(sh)) + rd(s2)
            if val > M64:
                raise Overflow(ln)
            wr(d, val)
        elif op == "v_lshrrev_b64":
            d, sh, s = args
            wr(d, rd(s) >> rd(sh))
        elif op == "v_alignbit_b32":
            # D = ({S0, S1} >> S2[4:0]) & 0xffffffff; the generator uses it as "column carry fits 32 bits":
            # a carry that does not fit would be silently truncated, so that is an overflow here
            d, hi, lo, sh = args
            val = ((rd(hi) << 32) | rd(lo)) >> (rd(sh) & 31)
            if val > M32:
                raise Overflow(ln)
            wr(d, val)
        elif op == "v_lshlrev_b32":
            d, sh, s = args
            val = rd(s) << rd(sh)
            if val > M32:
                raise Overflow(ln)
            wr(d, val)
        elif op == "v_and_b32":
            d, s0, s1 = args
            wr(d, rd(s0) & rd(s1))
        elif op in ("v_mov_b32", "s_mov_b32"):
            d, s = args
            wr(d, rd(s))
        elif op == "v_mul_lo_u32":
            d, s0, s1 = args
            wr(d, (rd(s0) * rd(s1)) & M32)
        elif op == "v_sub_u32":
            d, s0, s1 = args
            wr(d, (rd(s0) - rd(s1)) & M32)
        else:
            raise AssertionError("instruction not modelled: " + ln)
    return a


def value(limbs, W):
    return sum(v << (W * i) for i, v in enumerate(limbs))


def limbs_of(v, W, nl):
    return [(v >> (W * i)) & ((1 << W) - 1) for i in range(nl)]


@pytest.fixture(scope="module")
def asm():
    return parse_header()


@pytest.fixture(scope="module")
def moduli():
    with open(os.path.join(ROOT, "tests", "golden", "params.json")) as f:
        P = json.load(f)
    return [int(P[name]["modulus"]) for name in FIELD_IDS]


def adversarial(p, W, nl, mult, rng):
    """limb vectors at the edge of what the statement may be given: values below mult * p with as many
    limbs at their maximum as that allows"""
    full = (1 << W) - 1
    top = min(full, max(((mult * p) >> (W * (nl - 1))) - 1, 0))
    pats = [[full] * (nl - 1) + [top], [full] * (nl - 1) + [0], [0] * nl, [1] + [0] * (nl - 1),
            [full if i % 2 else 0 for i in range(nl - 1)] + [top], limbs_of(p - 1, W, nl), limbs_of(p, W, nl),
            limbs_of(mult * p - 1, W, nl)]
    for _ in range(6):
        pats.append([rng.randrange(full + 1) for _ in range(nl - 1)] + [rng.randrange(top + 1)])
    assert all(value(v, W) < mult * p for v in pats)
    return pats


def test_every_layout_is_present(asm):
    assert sorted(asm) == sorted([(f, 29) for f in range(7)] + [(0, 30), (1, 30)])


@pytest.mark.parametrize("key", [(f, 29) for f in range(7)] + [(0, 30), (1, 30)])
def test_generated_products_on_adversarial_limbs(asm, moduli, key):
    fid, W = key
    e, p = asm[key], moduli[fid]
    nl = e["NL"]
    Rinv = pow(1 << (W * nl), -1, p)
    rng = random.Random(fid * 100 + W)
    # contract (mont29.h): a product of inputs < A p and < B p with A B <= H = R'/p is below 2p; on 30-bit
    # limbs additionally a < 16 p (squaring) / b < 16 p (multiplication)
    H = (1 << (W * nl)) // p
    import math
    A = min(math.isqrt(H), 16 if W >= 30 else 1 << 12)
    lines, outs, ins = e["sqr"]
    for a in adversarial(p, W, nl, A, rng):
        r = run(lines, outs, ins, a)
        assert all(0 <= v < (1 << W) for v in r)
        assert value(r, W) % p == value(a, W) ** 2 * Rinv % p
        assert value(r, W) < 2 * p
    B = min(math.isqrt(H), 16)
    A = min(H // B, 1 << 20)       # `a` may be as large as the linear layer's values: A B <= H is all it needs
    lines, outs, ins = e["mul"]
    for a in adversarial(p, W, nl, A, rng):
        for b in adversarial(p, W, nl, B, rng)[:6]:
            r = run(lines, outs, ins, a, b)
            assert all(0 <= v < (1 << W) for v in r)
            assert value(r, W) % p == value(a, W) * value(b, W) * Rinv % p
            assert value(r, W) < 2 * p


def test_the_interpreter_notices_an_overflow():
    """sanity of the checker itself: five 32x32-bit products of all ones do not fit a 64-bit accumulator"""
    lines = ["v_mov_b32 v100, 0", "v_mov_b32 v101, 0"] + ["v_mad_u64_u32 v[100:101], vcc, %0, %0, v[100:101]"] * 5
    outs = [("a", "0")]
    assert run(lines[:3], outs, [], [M32]) == [M32]
    with pytest.raises(Overflow):
        run(lines, outs, [], [M32])


# ---- the wave-cooperative products (AsmCoop: one element per wavefront; AsmCoop4: one per 16-lane DPP row) ------------
# A 64-lane interpreter of the handful of instructions those statements use, DPP controls included
# (row_newbcast:i = lane i of the own 16-lane row; row_shl:1 with bound_ctrl = lane + 1 of the own row, 0 beyond it),
# with the same overflow detection.  What cannot be modelled here is timing (the s_nop hazard slots): that is what the
# GPU parity test with both kernels forced is for.

def parse_coop(struct):
    text = open(HDR).read()
    out = {}
    for m in re.finditer(r"template <> struct %s<(\d+)> \{(.*?)\n\};" % struct, text, re.S):
        fn = re.search(r"asm volatile\((.*?)\);", m.group(2), re.S).group(1)
        out[int(m.group(1))] = re.findall(r'"([^"]+?)\\n\\t"', fn)
    return out


def run_coop(lines, a, b, pl, sh, mask=None):
    """a, b, pl, sh: 64-entry lists (per lane); returns the 64 column sums t = (hi << 32) | lo"""
    L = 64
    v, sreg, vcc = {}, {}, [0] * L
    opnd = {"%2": a, "%3": b, "%4": pl, "%5": sh, "%6": [mask] * L if mask is not None else None}
    outv = {}

    def vec(tok):
        tok = tok.strip()
        if tok in opnd:
            return list(opnd[tok])
        if tok in ("%0", "%1"):
            return list(outv[tok])
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            lo, hi = v["v%s" % m.group(1)], v["v%d" % (int(m.group(1)) + 1)]
            return [l | (h << 32) for l, h in zip(lo, hi)]
        if tok.startswith("v"):
            return list(v[tok])
        if tok.startswith("s"):
            return [sreg[tok]] * L
        return [int(tok, 0) & M64] * L

    def put(tok, vals):
        tok = tok.strip()
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            v["v%s" % m.group(1)] = [x & M32 for x in vals]
            v["v%d" % (int(m.group(1)) + 1)] = [(x >> 32) & M32 for x in vals]
        elif tok in ("%0", "%1"):
            outv[tok] = [x & M32 for x in vals]
        else:
            v[tok] = [x & M32 for x in vals]

    def dpp(vals, ctrl):
        m = re.search(r"row_newbcast:(\d+)", ctrl)
        if m:
            return [vals[(l & ~15) + int(m.group(1))] for l in range(L)]
        assert "row_shl:1" in ctrl and "bound_ctrl:1" in ctrl, ctrl
        return [vals[l + 1] if (l & 15) != 15 else 0 for l in range(L)]

    for ln in lines:
        if ln.startswith(".p2align") or ln.startswith("v_nop"):
            continue
        op, _, rest = ln.partition(" ")
        op = re.sub(r"_e(32|64)$", "", op)
        ctrl = ""
        m = re.search(r"\s(row_[a-z_]+:.*)$", rest)
        if m:
            ctrl, rest = m.group(1), rest[:m.start()]
        args = [x.strip() for x in re.split(r",\s*(?![^\[]*\])", rest)] if rest.strip() else []
        if op == "s_nop":
            continue
        if op == "s_mov_b32":
            sreg[args[0]] = int(args[1], 0)
        elif op == "v_readlane_b32":
            sreg[args[0]] = vec(args[1])[int(args[2])]
        elif op == "s_mul_i32":
            sreg[args[0]] = (sreg[args[1]] * int(args[2], 0)) & M32
        elif op == "s_sub_i32":
            sreg[args[0]] = (int(args[1], 0) - sreg[args[2]]) & M32
        elif op == "s_and_b32":
            sreg[args[0]] = sreg[args[1]] & int(args[2], 0)
        elif op in ("v_mov_b32", "v_mov_b32_dpp"):
            src = vec(args[1])
            put(args[0], dpp(src, ctrl) if ctrl else src)
        elif op == "v_mad_u64_u32":
            s0, s1, s2 = vec(args[2]), vec(args[3]), vec(args[4])
            val = [x * y + z for x, y, z in zip(s0, s1, s2)]
            if max(val) > M64:
                raise Overflow(ln)
            put(args[0], val)
        elif op == "v_sub_u32":
            put(args[0], [(x - y) & M32 for x, y in zip(vec(args[1]), vec(args[2]))])
        elif op == "v_mul_lo_u32":
            put(args[0], [(x * y) & M32 for x, y in zip(vec(args[1]), vec(args[2]))])
        elif op == "v_and_b32_dpp":
            put(args[0], [x & y for x, y in zip(dpp(vec(args[1]), ctrl), vec(args[2]))])
        elif op == "v_lshrrev_b64":
            put(args[0], [x >> (s & 63) for s, x in zip(vec(args[1]), vec(args[2]))])
        elif op == "v_add_co_u32_dpp":
            s = [x + y for x, y in zip(dpp(vec(args[2]), ctrl), vec(args[3]))]
            vcc = [x >> 32 for x in s]
            put(args[0], s)
        elif op == "v_addc_co_u32_dpp":
            s = [x + y + c for x, y, c in zip(dpp(vec(args[2]), ctrl), vec(args[3]), vcc)]
            if max(s) > M32:
                raise Overflow(ln)     # the column sums are dimensioned to stay below 2^63
            put(args[0], s)
        else:
            raise AssertionError("instruction not modelled: " + ln)
    return [lo | (hi << 32) for lo, hi in zip(outv["%0"], outv["%1"])]


@pytest.mark.parametrize("struct,rows", [("AsmCoop", 1), ("AsmCoop4", 4)])
@pytest.mark.parametrize("fid", range(7))
def test_cooperative_products_on_adversarial_limbs(moduli, struct, rows, fid):
    """Each row's column sums must add up to a b / R' mod p (value level), stay below 2^63 (the per-lane shift by 63
    relies on it) and -- with four elements per wavefront -- no row may see anything of its neighbours."""
    W, p = 29, moduli[fid]
    nl = -(-(p.bit_length() + 6) // W)
    lines = parse_coop(struct)[fid]
    Rinv = pow(1 << (W * nl), -1, p)
    full = (1 << W) - 1 + (1 << 7)          # limbs as the previous product's settle_columns leaves them: < 2^29 + 2^7
    rng = random.Random(1000 + fid)
    H = (1 << (W * nl)) // p
    import math
    A = min(math.isqrt(H), 1 << 12)
    lpr = 64 // rows

    def operands():
        kind = rng.randrange(4)
        if kind == 0:
            return [full] * (nl - 1) + [min(full, (A * p) >> (W * (nl - 1)))]
        if kind == 1:
            return limbs_of(rng.randrange(A * p), W, nl)
        if kind == 2:
            return limbs_of(p - 1, W, nl)
        return [0] * nl
    pl_limbs = limbs_of(p, W, nl)
    for trial in range(6):
        a_rows = [operands() for _ in range(rows)]
        b_rows = [operands() for _ in range(rows)]
        lane = lambda vals_rows: [(vals_rows[l // lpr][l % lpr] if l // lpr < rows and l % lpr < nl else 0) for l in range(64)]
        a, b = lane(a_rows), lane(b_rows)
        pl = lane([pl_limbs] * rows)
        sh = [29 if l % lpr == 0 else 63 for l in range(64)]
        t = run_coop(lines, a, b, pl, sh, mask=(1 << W) - 1 if struct == "AsmCoop4" else None)
        for r in range(rows):
            cols = t[r * lpr:r * lpr + nl]
            assert all(c < (1 << 63) for c in cols)
            got = sum(c << (W * j) for j, c in enumerate(cols))
            av, bv = value(a_rows[r], W), value(b_rows[r], W)
            assert got % p == av * bv * Rinv % p, (struct, fid, trial, r)
            assert got < 2 * p + (av * bv >> (W * nl))      # (a b + m p) / R' with m < R'
            assert all(c == 0 for c in t[r * lpr + nl:(r + 1) * lpr])   # lanes beyond the element stay zero


def test_every_generated_statement_sits_on_the_8_byte_fetch_grid(asm):
    """tools/asm_grid.py: a wavefront that has its SIMD to itself pays ~1 cycle for every 8-byte instruction that
    straddles an 8-byte boundary, so every generated statement starts aligned and keeps its 8-byte instructions on the
    grid.  Checked with the assembler: each line's size is what the generator assumed, every 8-byte instruction starts
    at a multiple of 8."""
    import shutil
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import asm_grid
    mc = shutil.which("llvm-mc") or "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        pytest.skip("no llvm-mc")
    bodies = {}
    for key, e in asm.items():
        bodies[("sqr",) + key] = (e["sqr"][0], {"%%%d" % (e["NL"] + i): "s%d" % (30 + i) for i in range(e["NL"] + 1)})
        bodies[("mul",) + key] = (e["mul"][0], {})
    for struct in ("AsmCoop", "AsmCoop4"):
        for fid, lines in parse_coop(struct).items():
            bodies[(struct, fid)] = (lines, {})
    for key, (lines, sregs) in bodies.items():
        assert lines[0] == ".p2align 3", key
        text = []
        for ln in lines:
            ln = re.sub(r"%(\d+)", lambda m: sregs.get(m.group(0), "v%d" % (200 + int(m.group(1)))), ln)
            text.append(ln)
        out = subprocess.run([mc, "-triple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-show-encoding"],
                             input="\n".join(text) + "\n", capture_output=True, text=True, check=True).stdout
        sizes = [len(m.group(1).split(",")) for m in re.finditer(r"encoding: \[([^\]]*)\]", out)]
        insts = [ln for ln in lines if not ln.endswith(":") and not ln.startswith(".")]
        assert len(sizes) == len(insts), (key, len(sizes), len(insts))
        off = 0
        for ln, size in zip(insts, sizes):
            assert size == asm_grid.enc_size(ln), (key, ln, size)
            # (AsmCoop, the one-element scan of rounds 1-2, is kept for A/B and parity only: its SALU digit path leaves
            # 4-byte instructions that cannot be paired; the generator states how many instructions stay off the grid)
            assert size == 4 or off % 8 == 0 or key[0] == "AsmCoop", (key, ln, off)
            off += size
