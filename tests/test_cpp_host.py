"""The C++ host mirror of the reference's trait surface (anemoi-rust_amd/host/anemoi.hpp):
compiles on CPU; on the GPU box its reference-style test program runs every hasher KAT."""
import os
import subprocess

import pytest

from conftest import FIELD_IDS, ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "test_reference_style.cpp")
LIBDIR = os.path.join(ROOT, "anemoi-rust_amd", "lib")


def build(tmp_path):
    exe = str(tmp_path / "test_reference_style")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", SRC, "-o", exe, "-L" + LIBDIR, "-lanemoi_mi355x",
                           "-Wl,-rpath," + LIBDIR, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_host_mirror_compiles_and_links(tmp_path):
    assert os.path.exists(build(tmp_path))


@pytest.mark.gpu
def test_reference_style_cpp_tests(tmp_path, kats):
    exe = build(tmp_path)
    lines = []
    for name, k in kats.items():
        field, inst = name.split("/")
        fid, width = FIELD_IDS.index(field), 2 if inst == "anemoi_2_1" else 4
        for a, b in zip(k["hash_field"]["in"], k["hash_field"]["out"]):
            lines.append("%d %d hash_field %d 1 %s %s" % (fid, width, len(a), " ".join(a), b))
        for a, b in zip(k["hash_bytes"]["in_hex"], k["hash_bytes"]["out"]):
            lines.append("%d %d hash_bytes 1 1 %s %s" % (fid, width, a, b))
        for a, b in zip(k["jive"]["in"], k["jive"]["out"]):
            lines.append("%d %d jive %d %d %s %s" % (fid, width, len(a), len(b), " ".join(a), " ".join(b)))
        if width == 4:
            for a, b in zip(k["jive_k4"]["in"], k["jive_k4"]["out"]):
                lines.append("%d %d jive_k4 %d %d %s %s" % (fid, width, len(a), len(b), " ".join(a), " ".join(b)))
    # run-time instances (tests/golden/generic.json): `width` column = NUM_COLUMNS, inputs =
    # rounds, has_mds, ARK_C, ARK_D, [MDS], state; outputs = permutation(state)
    import json
    with open(os.path.join(ROOT, "tests", "golden", "generic.json")) as f:
        for v in json.load(f):
            ins = [str(v["num_rounds"]), "1" if v["mds"] else "0"] + v["ark_c"] + v["ark_d"] + (v["mds"] or []) + v["state"]
            lines.append("%d %d generic %d %d %s %s" % (FIELD_IDS.index(v["field"]), v["num_columns"], len(ins),
                                                        len(v["permutation"]), " ".join(ins), " ".join(v["permutation"])))
    path = tmp_path / "kats.txt"
    path.write_text("\n".join(lines) + "\n")
    out = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "%d vector lines, 0 failures" % len(lines) in out.stdout
