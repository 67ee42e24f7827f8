"""The 8-byte fetch grid of the generated gfx950 assembly (used by gen_coop2d_asm.py and gen_asm_mul.py).

A wavefront that has its SIMD to itself pays ~1 cycle extra for every 8-byte instruction that does not start on an 8-byte
boundary (tools/ubench/lone_wave_fetch.hip, profiles/r04/ubench_lone_wave_fetch.txt: a stream of 8-byte instructions
costs 4.3 cycles each aligned and 5.3 at offset 4; 4-byte instructions cost the same at any offset).  The generated
statements are mostly 8-byte instructions (VOP3, DPP, 32-bit literals), so they are laid out on an 8-byte grid: a
statement starts aligned (.p2align 3), and every run of 4-byte instructions between two 8-byte ones is made even by
giving one of them its 8-byte encoding (_e64; `s_nop k` becomes two s_nop with the same total wait).  Encodings
change, issue slots and semantics do not.  tests/test_coop2d_model.py and tests/test_asm_model.py check the result
against the assembler (llvm-mc): sizes as assumed here, every 8-byte instruction on the grid.
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


def align8(lines):
    """-> (lines on the 8-byte grid, number of padding instructions that had to be INSERTED (each costs an issue slot))"""
    out, run, off, inserted = [".p2align 3"], [], 0, 0   # run: indices (in out) of the 4-byte instructions since the last 8-byte one
    for ln in lines:
        size = enc_size(ln)
        if size == 8 and off % 8:
            for i in reversed(run):
                op, _, rest = out[i].partition(" ")
                lit = enc_size(out[i]) == 4
                if op in E32_PROMOTABLE and lit:
                    out[i] = op + "_e64 " + rest
                    break
                if op == "s_nop" and int(rest) >= 1:
                    out[i:i + 1] = ["s_nop %d" % (int(rest) - 1), "s_nop 0"]
                    break
                if op == "s_nop":
                    out[i] = "v_nop_e64"
                    break
            else:
                out.append("s_nop 0")
                inserted += 1
            off += 4
        out.append(ln)
        off += size
        if size == 8:
            run = []
        elif size == 4:
            run.append(len(out) - 1)
    return out, inserted
