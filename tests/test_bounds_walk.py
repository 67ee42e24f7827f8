"""The lazy-reduction bounds of every arithmetic in the library, machine-checked on the code the kernels are made of.

tests/cpp/bounds_walk/bounds_walk.cpp (the WITNESS) compiles the kernel bodies of anemoi-rust_amd/csrc for the host over
arithmetics that carry an upper bound of every value -- `ArithFor<FIELD>` and `CoopArith<F, LPI>::type` are swapped,
nothing else: permutation, flystel, mds_layer, mds_pair, mds_cols, the sponge absorb loops and their segment carry,
k_merkle_climb's feed-forward, the Jive sums, the ABI conversions and coop_permutation / coop_flystel are the product's
own templates -- and runs every kernel family of every field.  tools/bounds_walk.py (the JUDGE) re-derives every
distinct step with Python integers from tests/golden/params.json and checks each operation's preconditions.  Here:

  * the walk is green for 7 fields x {lane-private, scan, two-row fold} x both widths x every kernel family, and the
    families it must cover are all there;
  * it goes RED under mutations: a smaller subtraction pad in a mutated copy of field_consts_gen.h, a `settle` turned
    into a no-op, a witness that misreports a result;
  * the largest operands a lane-private product ever meets are run through the instruction-level interpreter of the
    GENERATED assembly (tests/test_asm_model.py) with every limb at its maximum: 64-bit column sums hold AT those bounds;
  * anemoi-rust_amd/csrc/BOUNDS.md -- the table the comments in anemoi_perm.h / anemoi_generic.h point to -- is what the
    walk prints today.
"""
import os
import re
import shutil
import sys

import pytest

from conftest import FIELD_IDS, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import bounds_walk as BW  # noqa: E402

HDR = os.path.join(ROOT, "anemoi-rust_amd", "csrc", "field_consts_gen.h")
TABLE = os.path.join(ROOT, "anemoi-rust_amd", "csrc", "BOUNDS.md")


@pytest.fixture(scope="module")
def logs():
    return BW.run(BW.build())


def test_every_step_of_every_kernel_family_is_allowed(logs, params):
    bad = BW.judge(logs, params)
    assert not bad, "\n".join(bad[:30])
    assert sorted(logs) == list(range(7))
    for f, log in logs.items():
        assert sum(s["n"] for s in log.steps) > 1_000_000, "the walk of %s is suspiciously short" % FIELD_IDS[f]
        # layouts as shipped: 30-bit lane limbs for the 381/377-bit fields, 29-bit scan limbs everywhere
        assert log.layout["lane"].W == (30 if f < 2 else 29) and log.layout["scan"].W == 29 and log.layout["fold"].W in (27, 28)


LANE_FAMILIES = ["k_permutation<2>", "k_permutation<2,sbox_only>", "k_jive<2,2>", "k_exp_alpha", "k_exp_inv_alpha",
                 "k_sponge<2,bytes> whole", "k_sponge<2,bytes> first segment", "k_sponge<2,bytes> last segment",
                 "k_sponge<2,elements> whole", "k_sponge<2,elements> first segment", "k_sponge<2,elements> last segment",
                 "k_sponge_ragged<bytes>", "k_sponge_ragged<elements>", "k_merkle_climb depth 3", "k_permutation<4> (one state per lane)",
                 "k_jive<4,2> (one state per lane)", "k_jive<4,4> (one state per lane)", "k_permutation_pair",
                 "k_permutation_pair<sbox_only>", "k_jive_pair<2>", "k_jive_pair<4>", "k_sponge_pair<bytes> whole",
                 "k_sponge_pair<bytes> first segment", "k_sponge_pair<bytes> last segment", "k_sponge_pair<elements> whole",
                 "k_sponge_pair<elements> first segment", "k_sponge_pair<elements> last segment", "k_sponge_ragged_pair<bytes>",
                 "k_sponge_ragged_pair<elements>"]
COOP_FAMILIES = ["k_jive2_coop<%s>", "k_permutation_coop<2,%s>", "k_merkle_climb_coop<%s> depth 3", "k_jive4_coop<2,%s>",
                 "k_jive4_coop<4,%s>", "k_permutation_coop<4,%s>"]
for _w in (2, 4):
    COOP_FAMILIES += ["k_sponge_coop<%d,bytes,%%s> %s" % (_w, part) for part in ("whole", "first segment", "last segment")]
    COOP_FAMILIES += ["k_sponge_coop<%d,elements,%%s> %s" % (_w, part) for part in ("whole", "last segment")]
    COOP_FAMILIES += ["k_sponge_ragged_coop<%d,%s,%%s>" % (_w, kind) for kind in ("bytes", "elements")]   # two messages of different lengths in one wavefront


def test_the_walk_covers_every_kernel_family(logs):
    want = ["lane " + c for c in LANE_FAMILIES] + ["lane run-time instance, NUM_COLUMNS = %d" % c for c in range(1, 17)]
    want += ["scan " + c % "16" for c in COOP_FAMILIES] + ["fold " + c % "32" for c in COOP_FAMILIES]
    want += ["fold k_sponge_ragged_coop<4,%s,32> short message" % kind for kind in ("bytes", "elements")]   # (the 4-3 fold kernel holds one message per wavefront)
    for f, log in logs.items():
        assert sorted(log.cases) == sorted(want), FIELD_IDS[f]
        # ... and every arithmetic statement of the headers was reached on every field's walk where it exists
        sites = {s["site"] for s in log.sites}
        for must in ("anemoi_perm.h", "anemoi_kernels.h", "anemoi_generic.h", "anemoi_coop_kernels.h"):
            assert any(s.startswith(must) for s in sites), (FIELD_IDS[f], must)


def test_the_judge_rejects_a_witness_that_misreports(logs, params):
    log = logs[4]
    s = dict(next(x for x in log.steps if x["op"] == "sqr" and x["arith"] == "lane"))
    assert BW.judge_step(4, log, s) == []
    s["out"] -= 1 << (s["out"].bit_length() - 20)
    assert any("disagree" in w for w in BW.judge_step(4, log, s))
    s = dict(next(x for x in log.steps if x["op"] == "sub" and x["arith"] == "lane"))
    s["b"] = 5 * log.layout["lane"].p                       # Jubjub pads with 4 p
    assert any("above the pad" in w for w in BW.judge_step(4, log, s))
    s = dict(next(x for x in log.steps if x["op"] == "add" and x["arith"] == "scan"))
    s["a"], s["out"] = log.layout["scan"].R, BW.round_up(log.layout["scan"].R + s["b"])
    assert any("not below R'" in w for w in BW.judge_step(4, log, s))
    s = dict(op="raw", arith="lane", a=1, b=0, out=1, n=1, first="x|y")   # limbs written around the arithmetic
    assert BW.judge_step(4, log, s)


def mutated_header(tmp, field, struct, sub_k):
    """a copy of field_consts_gen.h in which `struct <struct>` of FieldC<field> pads its subtractions with sub_k * p"""
    text = open(HDR).read()
    a = text.index("template <> struct FieldC<%d> {" % field)
    b = text.index("struct %s {" % struct, a)
    m = re.compile(r"static constexpr int W = (\d+), NL = (\d+), kSubK = (\d+);").search(text, b)
    W, NL = int(m.group(1)), int(m.group(2))
    mp = re.compile(r"static constexpr uint32_t P\[\d+\] = \{([^}]*)\};").search(text, b)
    p = sum(int(x.rstrip("u"), 16) << (W * i) for i, x in enumerate(mp.group(1).split(",")))
    q = [((sub_k * p) >> (W * i)) & ((1 << W) - 1) for i in range(NL)]
    kp = [q[0] + (1 << W)] + [q[i] + (1 << W) - 1 for i in range(1, NL - 1)] + [q[NL - 1] - 1]   # tools/gen_params.py's padding
    assert sum(v << (W * i) for i, v in enumerate(kp)) == sub_k * p
    mk = re.compile(r"static constexpr uint32_t KP\[\d+\] = \{([^}]*)\};").search(text, b)
    text = text[:mk.start(1)] + ",".join("0x%08xu" % v for v in kp) + text[mk.end(1):]
    # a whole copy of csrc: the kernels include "field_consts_gen.h" by quotes, i.e. from their own directory first
    shutil.copytree(os.path.dirname(HDR), tmp, ignore=shutil.ignore_patterns("*.hip", "*asm_gen*"))
    with open(os.path.join(tmp, "field_consts_gen.h"), "w") as f:
        f.write(text)
    return tmp


def test_a_smaller_pad_turns_the_walk_red(tmp_path, params):
    """BLS12-381 on 30-bit lane limbs pads with 8 p and subtracts g y^2 = 2 x (a square <= 1.00x p): the walk shows that
    even 4 p would do (the comments' "u < 4" was an over-estimate), a pad of 2 p does not cover the S-box's subtractions"""
    csrc = mutated_header(str(tmp_path / "csrc"), 0, "R30", 2)
    out = str(tmp_path / "bin")
    logs = BW.run(BW.build(out_dir=out, csrc=csrc, fields=[0]), fields=[0])
    bad = BW.judge(logs, params)
    assert bad and all("bls12_381 lane" in b for b in bad), bad[:5]
    assert any("subtrahend above the pad" in b and "anemoi_perm.h" in b for b in bad), bad[:5]
    shutil.rmtree(out)


def test_a_removed_settle_turns_the_walk_red(tmp_path, params):
    """Jubjub has the least headroom (H = 70.7): without the one settle per state element and round the sums outgrow
    what a product may be given"""
    out = str(tmp_path / "bin")
    logs = BW.run(BW.build(out_dir=out, extra="-DWALK_MUTATE_NO_SETTLE", fields=[4]), fields=[4])
    bad = BW.judge(logs, params)
    assert any("jubjub lane" in b and ("not below 2p" in b or "not below R'" in b) for b in bad), bad[:5]
    shutil.rmtree(out)


def test_generated_assembly_at_the_walked_bounds(logs):
    """the products' 64-bit column sums AT the largest operands any lane-private statement meets, every limb at its
    maximum, on the instruction-level interpreter of the generated squaring / multiplication"""
    import test_asm_model as AM
    asm = AM.parse_header()
    for f, log in logs.items():
        L = log.layout["lane"]
        e = asm[(f, L.W)]
        assert e["NL"] == L.NL
        full = (1 << L.W) - 1
        Rinv = pow(L.R, -1, L.p)

        def worst(bound):   # the largest limb vector a value <= bound can be: lower limbs full, top limb the bound's
            assert bound < L.R
            return [full] * (L.NL - 1) + [bound >> (L.W * (L.NL - 1))]

        sq = max((s["a"] for s in log.sites if s["arith"] == "lane" and s["op"] == "sqr"), default=0)
        a = worst(sq)
        r = AM.run(*e["sqr"], a)
        assert all(0 <= v <= full for v in r) and AM.value(r, L.W) % L.p == AM.value(a, L.W) ** 2 * Rinv % L.p, FIELD_IDS[f]
        pairs = {(s["a"], s["b"]) for s in log.sites
                 if s["arith"] == "lane" and s["op"] in BW.PRODUCT_OPS + ("to_abi",) and s["op"] != "sqr"}
        assert pairs
        for am, bm in pairs:
            a, b = worst(am), worst(bm)
            r = AM.run(*e["mul"], a, b)
            assert all(0 <= v <= full for v in r), FIELD_IDS[f]
            assert AM.value(r, L.W) % L.p == AM.value(a, L.W) * AM.value(b, L.W) * Rinv % L.p, FIELD_IDS[f]


def test_the_committed_table_is_current(logs):
    assert open(TABLE).read() == BW.table(logs), "run `python tools/bounds_walk.py --table anemoi-rust_amd/csrc/BOUNDS.md`"
