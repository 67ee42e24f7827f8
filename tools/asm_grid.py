"""The 8-byte fetch grid of the generated gfx950 assembly (used by gen_coop2d_asm.py and gen_asm_mul.py).

A wavefront that has its SIMD to itself pays ~1 cycle extra for every 8-byte instruction that does not start on an 8-byte
boundary (tools/ubench/lone_wave_fetch.hip, profiles/r04/ubench_lone_wave_fetch.txt: a stream of 8-byte instructions
costs 4.3 cycles each aligned and 5.3 at offset 4; 4-byte instructions cost the same at any offset).  The generated
statements are mostly 8-byte instructions (VOP3, DPP, 32-bit literals), so they are laid out on an 8-byte grid: a
statement starts aligned (.p2align 3), and the encodings of its 4-byte instructions are chosen -- a VOP1 / VOP2
instruction may take its 8-byte _e64 form, `s_nop k` may become two s_nop with the same total wait -- so that as few
8-byte instructions as possible straddle a boundary (a two-state dynamic programme over the text; for every shipped
statement but the digit-serial scans the minimum is zero).  Encodings change; issue slots and semantics do not.

What is NOT used as padding, and why (same microbenchmark, and the scan kernel at 4 096 / 8 192 items):
  * an SALU instruction with a literal (8 bytes) in a hazard gap: behind a VALU instruction that writes an SGPR
    (v_mad_u64_u32 writes vcc) it waits for that write -- +15 cycles per filler;
  * v_nop_e64 where wavefronts may share a SIMD: it occupies the vector ALU the other wavefront could use (12-23 % on
    the scan kernel at one / two wavefronts per SIMD); allowed (`vnop=True`) only for the two-row fold statements, which
    run at <= one wavefront per SIMD, where it costs exactly what the s_nop 0 it replaces costs.

tests/test_coop2d_model.py and tests/test_asm_model.py check the result against the assembler (llvm-mc): sizes as
assumed here, the 8-byte instructions of every statement on the grid (the scans: at most the count stated there).
"""
import re

# VOP1 / VOP2 instructions of the generators whose _e64 (VOP3) form means the same
E32_PROMOTABLE = ("v_add_u32", "v_sub_u32", "v_lshrrev_b32", "v_lshlrev_b32", "v_mov_b32", "v_and_b32",
                  "v_permlane16_swap_b32", "v_permlane32_swap_b32")
# instructions that only exist as (or are written in) an 8-byte encoding
ALWAYS_8 = ("v_mad_u64_u32", "v_lshl_add_u64", "v_alignbit_b32", "v_lshrrev_b64", "v_mul_lo_u32", "v_mul_hi_u32",
            "v_readlane_b32", "v_add3_u32", "v_cndmask_b32_e64")


def enc_size(line):
    """encoded bytes of one line of generated text"""
    if line.endswith(":") or line.startswith("."):
        return 0
    op, _, rest = line.partition(" ")
    if op == "v_nop":
        return 4
    if op.endswith("_dpp") or op.endswith("_e64") or op in ALWAYS_8:
        return 8
    args = [a.strip() for a in rest.split(",")]
    for a in args[(0 if op.startswith("s_cmp") or op.startswith("s_cbranch") or op in ("s_nop", "s_branch") else 1):]:
        if re.fullmatch(r"-?(0x[0-9a-fA-F]+|\d+)", a) and not -16 <= int(a, 0) <= 64:
            return 8              # a 32-bit literal follows the instruction word
    return 4


def _variants(line, vnop):
    """the ways to write `line` at no cost in issue slots: [(lines, bytes)]"""
    size = enc_size(line)
    out = [([line], size)]
    if size == 4:
        op, _, rest = line.partition(" ")
        if op in E32_PROMOTABLE:
            out.append(([op + "_e64 " + rest], 8))
        elif op == "s_nop" and int(rest) >= 1:
            out.append((["s_nop %d" % (int(rest) - 1), "s_nop 0"], 8))
        elif op == "s_nop" and vnop:
            out.append((["v_nop_e64"], 8))
    return out


def align8(lines, vnop=False):
    """-> (the statement on the 8-byte grid, number of 8-byte instructions left straddling a boundary)"""
    INF = 1 << 30
    cost = {0: 0, 4: INF}                       # parity of the current offset -> fewest straddlers so far
    back = []                                   # per line: {parity after: (parity before, variant index)}
    for ln in lines:
        new, step = {0: INF, 4: INF}, {}
        for p in (0, 4):
            if cost[p] >= INF:
                continue
            for vi, (_, size) in enumerate(_variants(ln, vnop)):
                q = (p + size) % 8
                c = cost[p] + (1 if size == 8 and p else 0)
                if c < new[q]:
                    new[q], step[q] = c, (p, vi)
        cost = new
        back.append(step)
    p = min(cost, key=cost.get)
    total = cost[p]
    chosen = []
    for ln, step in zip(reversed(lines), reversed(back)):
        p, vi = step[p]
        chosen.append(_variants(ln, vnop)[vi][0])
    out = [".p2align 3"]
    for c in reversed(chosen):
        out += c
    return out, total
