"""The register budget of the shipped kernels is a design point (13-limb kernels at
<= 168 VGPRs = 3 waves per SIMD, 9-limb kernels at <= 128 = 4 waves per SIMD, no scratch) -- and it is a
property of the BUILT code objects, which no source-level test sees: round 2 shipped every hot BLS12-377
kernel at 171-181 VGPRs (2 waves per SIMD) because hipcc hoisted two 13-limb constants out of the round loop.
This test reads the gfx950 code objects inside the library (tools/kernel_resources.py) and fails when a kernel
leaves its budget, so the next edit cannot do that silently.  CPU-only: nothing is launched."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(kr.DEFAULT_LIB), reason="library not built")


@pytest.fixture(scope="module")
def rows():
    return kr.collect()


def test_every_kernel_of_every_field_is_in_the_table(rows):
    names = {(r["kernel"].split("<")[0], r["field"]) for r in rows}
    for f in kr.FIELDS:
        for k in ("k_jive", "k_jive_pair", "k_sponge", "k_sponge_pair", "k_sponge_ragged", "k_sponge_ragged_pair",
                  "k_permutation", "k_permutation_pair", "k_merkle_climb", "k_jive2_coop", "k_jive4_coop", "k_sponge_coop", "k_sponge_ragged_coop", "k_permutation_coop", "k_merkle_climb_coop", "k_mont_convert",
                  "k_permutation_cols", "k_jive_cols", "k_sponge_cols", "k_exp_alpha"):
            assert (k, f) in names, (k, f)


def test_no_kernel_uses_scratch_and_hot_kernels_stay_in_their_register_budget(rows):
    bad = kr.check(rows)
    assert not bad, "\n".join(bad)


def test_headline_kernel_runs_three_waves_per_simd(rows):
    (r,) = [r for r in rows if r["kernel"] == "k_jive<0, 2, 2>"]
    assert r["vgpr"] + r["agpr"] <= 168 and r["waves_per_simd_regs"] == 3
    assert r["waves_per_cu"] == 12   # 12 288 B of LDS per wavefront: 13 would fit the LDS, the registers say 12
    assert r["scratch"] == 0


def test_committed_table_matches_the_built_library(rows):
    """profiles/r06/kernel_resources.csv is what the round's documents quote: it must describe this build"""
    path = os.path.join(ROOT, "profiles", "r06", "kernel_resources.csv")
    if not os.path.exists(path):
        pytest.skip("table not committed yet")
    import csv
    with open(path) as f:
        committed = {(r["kernel"], r["field"]): r for r in csv.DictReader(f)}
    for r in rows:
        c = committed.get((r["kernel"], r["field"]))
        assert c is not None, "kernel %s missing from the committed table: rerun tools/kernel_resources.py --csv" % r["kernel"]
        assert int(c["vgpr"]) == r["vgpr"] and int(c["scratch"]) == r["scratch"], (
            "%s: committed %s VGPRs, built %d: rerun tools/kernel_resources.py --csv %s"
            % (r["kernel"], c["vgpr"], r["vgpr"], os.path.relpath(path, ROOT)))
