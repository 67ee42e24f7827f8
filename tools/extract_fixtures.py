#!/usr/bin/env python3
"""Text-extract the reference's known-answer vectors and instance parameters.

Reads the reference crate's `.rs` files AS TEXT (nothing is compiled or run) and
writes pure data:

  tests/golden/kats.json     every KAT of the reference's unit tests, as decimal
                             canonical integers:
                               test_sbox            src/<f>/anemoi_*/mod.rs:68
                               test_anemoi_hash     src/<f>/anemoi_*/hasher.rs (hash_field)
                               test_anemoi_hash_bytes
                               test_anemoi_jive     (compress / compress_k(2) / merge; k=4 on 4-3)
  tests/golden/params.json   per-field ALPHA / INV_ALPHA / BETA / DELTA (src/<f>/sbox.rs:7-25),
                             per-instance sizes (anemoi_*/mod.rs:20-32) and round constants
                             C, D (anemoi_*/round_constants.rs), plus the field moduli.

The moduli are NOT in the reference (they live in the arkworks curve crates); they
are the well-known curve constants listed in SURVEY.md §8a and are validated here by
    ALPHA * INV_ALPHA == 1 (mod p-1)   and   BETA * DELTA == 1 (mod p).

Run from the repo root in the build container (needs /root/reference):
    python tools/extract_fixtures.py
"""
import json
import os
import re
import sys

REF = os.environ.get("ANEMOI_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# field dir -> (modulus, arkworks type, u64 limbs, byte chunk)
FIELDS = {
    "bls12_381": (0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab, "ark_bls12_381::Fq", 6),
    "bls12_377": (0x1ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001, "ark_bls12_377::Fq", 6),
    "bn_254": (0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47, "ark_bn254::Fq", 4),
    "ed_on_bls12_377": (0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001, "ark_bls12_377::Fr", 4),
    "jubjub": (0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001, "ark_bls12_381::Fr", 4),
    "pallas": (0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001, "ark_pallas::Fq", 4),
    "vesta": (0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001, "ark_pallas::Fr", 4),
}
WIDTHS = {"anemoi_2_1": 2, "anemoi_4_3": 4}

TOKEN = re.compile(
    r"""(?P<open>(?:vec!)?\[)|(?P<close>\])|(?P<zero>Felt::zero\(\))|(?P<one>Felt::one\(\))"""
    r"""|MontFp!\(\s*"(?P<num>\d+)"\s*,?\s*\)|;\s*(?P<rep>\d+)\s*(?=\])""",
    re.S,
)


def parse_array(text, pos):
    """Parse one (nested) Rust array literal starting at the first '[' at/after pos."""
    stack, root = [], None
    for m in TOKEN.finditer(text, pos):
        if m.group("open"):
            new = []
            if stack:
                stack[-1].append(new)
            stack.append(new)
        elif m.group("close"):
            done = stack.pop()
            if not stack:
                root = done
                return root, m.end()
        elif not stack:
            continue
        elif m.group("zero"):
            stack[-1].append(0)
        elif m.group("one"):
            stack[-1].append(1)
        elif m.group("num"):
            stack[-1].append(int(m.group("num")))
        elif m.group("rep"):
            cur = stack[-1]
            assert len(cur) == 1, "repeat syntax on a non-singleton"
            cur.extend([cur[0]] * (int(m.group("rep")) - 1))
    raise ValueError("unterminated array")


def after(text, needle, pos=0):
    i = text.index(needle, pos)
    return i + len(needle)


def read(*parts):
    with open(os.path.join(REF, "src", *parts)) as f:
        return f.read()


def const_u32(text, name):
    return int(re.search(r"const %s: u32 = (\d+);" % name, text).group(1))


def const_felt(text, name):
    return int(re.search(r'const %s: Felt\s*=\s*MontFp!\(\s*"(\d+)"\s*\)' % name, text, re.S).group(1))


def const_usize(text, name):
    return int(re.search(r"pub const %s: usize = (\d+);" % name, text).group(1))


def s(v):
    """ints -> decimal strings, recursively (JSON has no big ints)."""
    if isinstance(v, list):
        return [s(x) for x in v]
    return str(v)


def main():
    params, kats = {}, {}
    for field, (p, ark, limbs) in FIELDS.items():
        sbox = read(field, "sbox.rs")
        alpha, beta = const_u32(sbox, "ALPHA"), const_u32(sbox, "BETA")
        inv_alpha, delta = const_felt(sbox, "INV_ALPHA"), const_felt(sbox, "DELTA")
        assert alpha * inv_alpha % (p - 1) == 1, field
        assert beta * delta % p == 1, field
        fp = {
            "modulus": str(p), "arkworks_type": ark, "u64_limbs": limbs,
            "alpha": alpha, "inv_alpha": str(inv_alpha), "beta": beta, "delta": str(delta),
            "instances": {},
        }
        for inst, width in WIDTHS.items():
            mod = read(field, inst, "mod.rs")
            assert const_usize(mod, "STATE_WIDTH") == width
            rate, cols = const_usize(mod, "RATE_WIDTH"), const_usize(mod, "NUM_COLUMNS")
            rounds = const_usize(mod, "NUM_HASH_ROUNDS")
            rc = read(field, inst, "round_constants.rs")
            # start after the '=' so the type's own [Felt; N] bracket is skipped
            c, pos = parse_array(rc, after(rc, "=", after(rc, "const C:")))
            d, _ = parse_array(rc, after(rc, "=", after(rc, "const D:")))
            assert len(c) == len(d) == cols * rounds, (field, inst)
            assert all(0 <= v < p for v in c + d)
            fp["instances"][inst] = {
                "state_width": width, "rate_width": rate, "num_columns": cols,
                "num_rounds": rounds, "ark_c": s(c), "ark_d": s(d),
            }

            k = {}
            tb = mod[mod.index("fn test_sbox"):]
            sin, pos = parse_array(tb, after(tb, "let mut input ="))
            sout, _ = parse_array(tb, after(tb, "let output =", pos))
            assert len(sin) == len(sout) and all(len(v) == width for v in sin + sout)
            k["sbox"] = {"in": s(sin), "out": s(sout)}

            h = read(field, inst, "hasher.rs")
            chunk = int(re.search(r"bytes\.len\(\) % (\d+) == 0", h).group(1))
            fp["byte_chunk"] = chunk
            tb = h[h.index("fn test_anemoi_hash()"):h.index("fn test_anemoi_hash_bytes")]
            hin, pos = parse_array(tb, after(tb, "let input_data ="))
            hout, _ = parse_array(tb, after(tb, "let output_data =", pos))
            assert len(hin) == len(hout)
            k["hash_field"] = {"in": s(hin), "out": s([o[0] for o in hout])}

            tb = h[h.index("fn test_anemoi_hash_bytes"):h.index("fn test_anemoi_jive")]
            bin_, pos = parse_array(tb, after(tb, "let input_data ="))
            bout, _ = parse_array(tb, after(tb, "let output_data =", pos))
            # the test packs each element as its first `chunk` LE canonical bytes
            assert all(v < 256 ** chunk for e in bin_ for v in e)
            k["hash_bytes"] = {
                "in_elems": s(bin_),
                "in_hex": [b"".join(v.to_bytes(chunk, "little") for v in e).hex() for e in bin_],
                "out": s([o[0] for o in bout]),
            }

            tb = h[h.index("fn test_anemoi_jive"):]
            jin, pos = parse_array(tb, after(tb, "let input_data ="))
            jout, pos = parse_array(tb, after(tb, "let output_data =", pos))
            assert all(len(o) == cols for o in jout)
            k["jive"] = {"in": s(jin), "out": s(jout)}
            if width == 4:
                j4in, pos = parse_array(tb, after(tb, "let input_data =", pos))
                j4out, _ = parse_array(tb, after(tb, "let output_data =", pos))
                k["jive_k4"] = {"in": s(j4in), "out": s(j4out)}
            kats["%s/%s" % (field, inst)] = k
        params[field] = fp

    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    with open(os.path.join(ROOT, "tests", "golden", "params.json"), "w") as f:
        json.dump(params, f, indent=0, sort_keys=True)
    with open(os.path.join(ROOT, "tests", "golden", "kats.json"), "w") as f:
        json.dump(kats, f, indent=0, sort_keys=True)
    nk = sum(len(v["in"]) if "in" in v else len(v["in_elems"]) for k in kats.values() for v in k.values())
    print("instances: %d, KAT vectors: %d" % (len(kats), nk))


if __name__ == "__main__":
    sys.exit(main())
