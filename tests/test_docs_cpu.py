"""Document drift (round 4's review found a comment citing a tool that did not exist): every path under profiles/, tools/,
tests/, oracle/, include/ or the package that the documents, the header and the kernel sources' comments name must exist in
the tree, and the counts the documents quote must be the header's.  CPU-only."""
import os
import re

from conftest import ROOT

DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", "profiles/README.md", "include/anemoi_mi355x.h", "bench.py", "__graft_entry__.py"]
PATH = re.compile(r"(?<![\w/.-])((?:\.\./)?(?:profiles|tools|tests|oracle|include|integration|anemoi-rust_amd)/[\w./-]*\w\.(?:py|sh|md|txt|json|csv|hip|hpp|cpp|toml|rs|h|c)(?![\w]))")


def cited_paths(rel):
    text = open(os.path.join(ROOT, rel)).read()
    base = os.path.dirname(os.path.join(ROOT, rel))
    for m in PATH.finditer(text):
        p = m.group(1)
        yield p, os.path.normpath(os.path.join(base, p)) if p.startswith("../") else os.path.join(ROOT, p)


def test_every_path_the_documents_name_exists():
    missing = []
    for doc in DOCS:
        for shown, path in cited_paths(doc):
            if "rNN" in shown or "<" in shown or "gpurun_out" in shown:
                continue
            if not os.path.exists(path) and not os.path.exists(os.path.join(ROOT, "profiles", shown)):   # profiles/README.md names files relative to profiles/
                missing.append("%s: %s" % (doc, shown))
    assert not missing, "\n".join(sorted(set(missing)))


def test_every_profile_the_index_lists_exists():
    """profiles/README.md names its files relative to profiles/: `r05/host_api.txt`"""
    text = open(os.path.join(ROOT, "profiles", "README.md")).read()
    names = set(re.findall(r"`(r0\d/[\w./-]*\w\.(?:txt|json|csv|md))`", text))
    assert len(names) > 80
    missing = sorted(n for n in names if not os.path.exists(os.path.join(ROOT, "profiles", n)))
    assert not missing, missing


def test_kernel_source_comments_name_existing_files():
    missing = []
    csrc = os.path.join(ROOT, "anemoi-rust_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".h", ".hip")) or name.endswith("_gen.h"):
            continue
        for shown, path in cited_paths(os.path.join("anemoi-rust_amd", "csrc", name)):
            if not os.path.exists(path):
                missing.append("%s: %s" % (name, shown))
    assert not missing, "\n".join(sorted(set(missing)))


def test_quoted_counts_are_the_headers():
    header = open(os.path.join(ROOT, "include", "anemoi_mi355x.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    n = len(set(re.findall(r"\b(anemoi_[a-z0-9_]+)\s*\(", header)))
    for doc in ("DESIGN.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for m in re.finditer(r"(\d+) functions", text):
            assert int(m.group(1)) == n, "%s says %s functions, the header declares %d" % (doc, m.group(1), n)
    # the ABI version: DESIGN.md quotes "ABI <major>.<minor>", the header defines both, the library computes its answer from
    # the header's defines (the rule -- what moves which -- is the header's "ABI VERSION RULE" paragraph)
    raw = open(os.path.join(ROOT, "include", "anemoi_mi355x.h")).read()
    major = int(re.search(r"#define ANEMOI_ABI_MAJOR (\d+)", raw).group(1))
    minor = int(re.search(r"#define ANEMOI_ABI_MINOR (\d+)", raw).group(1))
    assert "ABI VERSION RULE" in raw
    abi = re.search(r"ABI (\d+)\.(\d+)", open(os.path.join(ROOT, "DESIGN.md")).read())
    assert abi and (int(abi.group(1)), int(abi.group(2))) == (major, minor)
    src = open(os.path.join(ROOT, "anemoi-rust_amd", "csrc", "capi.hip")).read()
    assert "anemoi_abi_version(void) { return 100 * ANEMOI_ABI_MAJOR + ANEMOI_ABI_MINOR; }" in src
    # ... and the header's own count of its functions, in the rule's history line
    assert ("(%d functions)" % n) in raw


# ---------------------------------------------------------------- committed profile summaries regenerate, and say what the documents say

def _profile_rounds():
    return sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d\d", d))


def _newest(name):
    for r in reversed(_profile_rounds()):
        p = os.path.join(ROOT, "profiles", r, name)
        if os.path.exists(p):
            return r, p
    raise AssertionError("no profiles/rNN/%s" % name)


def test_config_profile_summaries_are_sane():
    """Round 5's review, weak item 4: profiles/r04 and r05/pmc_configs.json said "config 3 = 1.91 ms, 166 x the flat rate" for
    two rounds (the summariser picked the profile tool's own small warm-up launch of the same kernel).  Every committed
    pmc_configs.json that carries the per-row `grid` (r04 on) must hold figures that can be true."""
    import json
    seen = 0
    for r in _profile_rounds():
        p = os.path.join(ROOT, "profiles", r, "pmc_configs.json")
        if not os.path.exists(p):
            continue
        d = json.load(open(p))
        s = d["summary"]
        if "cfg5_fraction_of_flat_rate" not in s:
            continue                                   # r03: the older layout, superseded
        seen += 1
        assert 0.85 < s["cfg3_fraction_of_flat_rate"] <= 1.02, (r, s["cfg3_fraction_of_flat_rate"])
        assert 250.0 < s["cfg3_ms"] < 450.0, (r, s["cfg3_ms"])
        assert 0.6 < s["cfg5_fraction_of_flat_rate"] <= 1.0, (r, s["cfg5_fraction_of_flat_rate"])
        assert 60.0 < s["cfg5_sum_of_levels_ms"] < 120.0, (r, s["cfg5_sum_of_levels_ms"])
        assert 25.0 < s["flat_jubjub_2_1_M_per_s"] < 40.0 and 18.0 < s["flat_bn254_4_3_M_per_s"] < 30.0, r
        cfg3 = [k for k in d["kernels"] if k["kernel"].startswith("k_sponge_pair<2, true>") and k["grid"] == 131072]
        assert len(cfg3) == 1 and abs(cfg3[0]["kernel_ms_min_stats_pass"] - s["cfg3_ms"]) < 1e-3, r   # the CONFIG's dispatch: 2^16 messages
        for k in d["kernels"]:
            t = k["traffic_over_algorithmic"]
            assert t is None or 0.95 < t < 400.0, (r, k["kernel"], k["grid"], t)          # (a lone wavefront reads 44 KB of constants for 192 bytes)
            if t is not None and k["wavefronts"] >= 2048:
                assert t < 1.5, (r, k["kernel"], k["grid"], t)                             # a full launch wastes nothing
            assert 0.5 < k["kernel_ms_min_stats_pass"] < 500.0 and 1.5 < k["clock_GHz"] < 2.6, (r, k["kernel"], k["grid"])
    assert seen >= 2


def test_design_quotes_the_newest_profiles():
    """DESIGN.md section 5 quotes config 3 / config 5 / the headline from the newest round's committed files, in sentences
    this test can read; a figure that drifts from its file fails here."""
    import json
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    r, p = _newest("pmc_configs.json")
    s = json.load(open(p))["summary"]
    m = re.search(r"`profiles/(r\d\d)/pmc_configs\.json`: config 3 (\d+\.\d) ms = (0\.\d+) of the flat rate; config 5 (\d+\.\d) ms as the sum of its "
                  r"levels = (0\.\d+) of the flat rate", text)
    assert m, "DESIGN.md does not quote profiles/%s/pmc_configs.json in the agreed sentence" % r
    assert m.group(1) == r, "DESIGN.md quotes %s, the newest summary is %s" % (m.group(1), r)
    assert abs(float(m.group(2)) - s["cfg3_ms"]) < 0.06 and abs(float(m.group(3)) - s["cfg3_fraction_of_flat_rate"]) < 0.0006
    assert abs(float(m.group(4)) - s["cfg5_sum_of_levels_ms"]) < 0.06 and abs(float(m.group(5)) - s["cfg5_fraction_of_flat_rate"]) < 0.0006
    r, p = _newest("bench_n1.json")
    line = json.loads([l for l in open(p).read().splitlines() if l.startswith("{")][-1])
    m = re.search(r"`profiles/(r\d\d)/bench_n1\.json`: (\d+\.\d\d) M/s, (\d+\.\d) ms per 2\^20 batch, (\d+\.\d) Mcycles", text)
    assert m and m.group(1) == r, "DESIGN.md does not quote profiles/%s/bench_n1.json in the agreed sentence" % r
    assert abs(float(m.group(2)) - line["value"] / 1e6) < 0.006
    assert abs(float(m.group(3)) - line["ms_per_step"]) < 0.06
    assert abs(float(m.group(4)) - line["alu"]["kernel_Mcycles_slowest_xcd"]) < 0.06


def test_design_quotes_the_collected_test_counts():
    """DESIGN.md section 4 says how many tests each suite holds: compared with what pytest collects (round 5 said 308 where
    the driver collected 309)."""
    import subprocess
    import sys
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    for marker, pat in (("gpu", r"`pytest -m gpu` \((\d+) tests"), ("not gpu", r"`-m \"not gpu\"` \((\d+) tests")):
        out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "--collect-only", "-q", "-m", marker],
                             capture_output=True, text=True, timeout=600, cwd=ROOT).stdout
        m = re.search(r"(\d+)/\d+ tests collected|(\d+) tests collected", out)
        assert m, out[-500:]
        n = int(m.group(1) or m.group(2))
        q = re.search(pat, text)
        assert q, "DESIGN.md section 4 does not state the %r count in the agreed form" % marker
        assert int(q.group(1)) == n, "DESIGN.md says %s %r tests, pytest collects %d" % (q.group(1), marker, n)
