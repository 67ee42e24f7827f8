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
