"""Facts about the kernel sources next to this package: a content hash (so measured profiles can say which
build they describe) and the multiply-add count of the headline kernel, read from the generated assembly.
Pure Python, no device access; used by bench.py and tools/summarize_profiles.py."""
import hashlib
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_ROOT, "csrc")


HOST_ONLY = ("capi.hip", "runtime.h", "host_logic.h")   # no device code: they cannot change a kernel


def csrc_sha256():
    """SHA-256 over the KERNEL sources and the Makefile (file names and contents, sorted): every csrc file
    except the host-only ones."""
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".hip")) and f not in HOST_ONLY)
    for path in [os.path.join(CSRC, f) for f in files] + [os.path.join(_ROOT, "Makefile")]:
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()


def bls12_381_limb_layout():
    """v_mad_u64_u32 / instruction counts per squaring and multiplication of the shipped BLS12-381 layout
    (the widest limbs listed in mont29_asm_gen.h = the default `Lane` layout)."""
    hdr = open(os.path.join(CSRC, "mont29_asm_gen.h")).read()
    best = None
    for m in re.finditer(r"// bls12_381, (\d+)-bit limbs: (\d+) limbs; squaring (\d+) instructions \((\d+) v_mad_u64_u32, "
                         r"\d+ split columns\), multiplication (\d+) \((\d+),", hdr):
        best = {"limb_bits": int(m.group(1)), "limbs": int(m.group(2)), "sqr_instr": int(m.group(3)),
                "sqr_mad": int(m.group(4)), "mul_instr": int(m.group(5)), "mul_mad": int(m.group(6))}
    def body(kind):   # the string macro holding the statement's instructions
        m = re.search(r"#define ANEMOI_ASM_%s_BODY_0_%d \\\n((?:[ \t]+\"[^\n]*\n)+)" % (kind, best["limb_bits"]), hdr)
        return m.group(1)
    wide = lambda s: sum(s.count(op) for op in ("v_lshrrev_b64", "v_lshl_add_u64", "v_mul_lo_u32"))
    best["sqr_wide"], best["mul_wide"] = wide(body("SQR")), wide(body("MUL"))
    return best


def _field0_array(hdr, name):
    blk = hdr[hdr.index("struct FieldC<0>"):]
    m = re.search(r"%s\[[^\]]*\]\s*=\s*\{([^}]*)\}" % name, blk)
    return [int(v, 0) for v in m.group(1).replace("\n", " ").split(",") if v.strip()]


def bls12_381_products_per_round():
    """(squarings, multiplications) one round of k_jive<bls12_381> executes: the S-box's exponentiation -- window 3
    plus the extra digits held in VGPRs when the field has them (field_consts_gen.h kXSched + the digits' build
    programmes), else the plain kW3Sched -- with x^2 and the table build, + its two y^2 + the two settle() products."""
    hdr = open(os.path.join(CSRC, "field_consts_gen.h")).read()
    blk = hdr[hdr.index("struct FieldC<0>"):]
    nx = int(re.search(r"kXDigits = (\d+)", blk).group(1))
    if nx:
        vals = _field0_array(hdr, "kXSched")
        ops, args = _field0_array(hdr, "kXProgOp"), _field0_array(hdr, "kXProgArg")
        build_sq = sum(a for o, a in zip(ops, args) if o == 1)
        build_mu = sum(1 for o in ops if o == 2)
    else:
        vals, build_sq, build_mu = _field0_array(hdr, "kW3Sched"), 0, 0
    pairs = list(zip(vals[0::2], vals[1::2]))
    sq = sum(s for s, _ in pairs) + 1 + build_sq + 2            # chain + x^2 + digit builds + the two y^2
    mu = sum(1 for _, op in pairs if op not in (255, 253)) + 3 + build_mu + 2   # chain + x^3, x^5, x^7 + builds + 2 settles
    return sq, mu


def bls12_381_mad_per_compression():
    lay = bls12_381_limb_layout()
    sq, mu = bls12_381_products_per_round()
    # 21 rounds + from_abi x 2 and to_abi x 1 (one Montgomery product each) + the final mds_layer's 2 settles
    return 21 * (sq * lay["sqr_mad"] + mu * lay["mul_mad"]) + 5 * lay["mul_mad"]


# ---- the same counts for any field / width (tools/bench_configs.py: alu_frac of every config) -------------------
FIELD_NAMES = ["bls12_381", "bls12_377", "bn_254", "ed_on_bls12_377", "jubjub", "pallas", "vesta"]


def _field_block(hdr, fid):
    a = hdr.index("struct FieldC<%d>" % fid)
    b = hdr.find("struct FieldC<%d>" % (fid + 1))
    return hdr[a:b if b > 0 else len(hdr)]


def _array(blk, name):
    m = re.search(r"%s\[[^\]]*\]\s*=\s*\{([^}]*)\}" % name, blk)
    return [int(v, 0) for v in m.group(1).replace("\n", " ").split(",") if v.strip()]


def limb_layout(fid):
    """instruction / v_mad_u64_u32 counts per squaring and multiplication of field `fid`'s lane-private layout"""
    name = FIELD_NAMES[fid]
    hdr = open(os.path.join(CSRC, "mont29_asm_gen.h")).read()
    best = None
    for m in re.finditer(r"// %s, (\d+)-bit limbs: (\d+) limbs; squaring (\d+) instructions \((\d+) v_mad_u64_u32, "
                         r"\d+ split columns\), multiplication (\d+) \((\d+)," % name, hdr):
        best = {"limb_bits": int(m.group(1)), "limbs": int(m.group(2)), "sqr_instr": int(m.group(3)),
                "sqr_mad": int(m.group(4)), "mul_instr": int(m.group(5)), "mul_mad": int(m.group(6))}
    return best


def is_tight(fid):
    """g * x is a Montgomery product (mont29.h mul_g) in the lane layout the kernels use"""
    blk = _field_block(open(os.path.join(CSRC, "field_consts_gen.h")).read(), fid)
    lay = "R30" if limb_layout(fid)["limb_bits"] == 30 else "R29"
    sub = blk[blk.index("struct %s" % lay):]
    return re.search(r"kTight = (true|false)", sub).group(1) == "true"


def flystel_products(fid):
    """(squarings, multiplications) of ONE Flystel S-box (anemoi_perm.h flystel): the exponentiation's schedule with
    x^2, the table build and the extra digits' build programmes, the two y^2, and g * y^2 twice where that is a product"""
    blk = _field_block(open(os.path.join(CSRC, "field_consts_gen.h")).read(), fid)
    nx = int(re.search(r"kXDigits = (\d+)", blk).group(1))
    if nx:
        vals = _array(blk, "kXSched")
        ops, args = _array(blk, "kXProgOp"), _array(blk, "kXProgArg")
        build_sq = sum(a for o, a in zip(ops, args) if o == 1)
        build_mu = sum(1 for o in ops if o == 2)
    else:
        vals, build_sq, build_mu = _array(blk, "kW3Sched"), 0, 0
    pairs = list(zip(vals[0::2], vals[1::2]))
    sq = sum(s for s, _ in pairs) + 1 + build_sq + 2
    mu = sum(1 for _, op in pairs if op not in (255, 253)) + 3 + build_mu + (2 if is_tight(fid) else 0)
    return sq, mu


def rounds(fid, width):
    blk = _field_block(open(os.path.join(CSRC, "field_consts_gen.h")).read(), fid)
    m = re.search(r"kRounds21 = (\d+), kRounds43 = (\d+)", blk)
    return int(m.group(1 if width == 2 else 2))


def mad_per_permutation(fid, width):
    """v_mad_u64_u32 per LANE-level permutation x the lanes a state occupies: width 2 = one lane (flystel + 2 settles
    per round); width 4 = a lane pair, each lane one flystel, 2 settles and (tight fields) 2 g-products per round"""
    lay = limb_layout(fid)
    sq, mu = flystel_products(fid)
    tight = is_tight(fid)
    r = rounds(fid, width)
    if width == 2:
        per_round = sq * lay["sqr_mad"] + (mu + 2) * lay["mul_mad"]
        return r * per_round + 2 * lay["mul_mad"]
    per_lane_round = sq * lay["sqr_mad"] + (mu + 2 + (2 if tight else 0)) * lay["mul_mad"]
    return 2 * (r * per_lane_round + (2 + (2 if tight else 0)) * lay["mul_mad"])


def mad_per_compression(fid, width):
    """Jive compression = permutation + ABI conversions (width elements in, width / 2 out for k = 2)"""
    return mad_per_permutation(fid, width) + (width + width // 2) * limb_layout(fid)["mul_mad"]
