"""The pure host side of the C-ABI (anemoi-rust_amd/csrc/host_logic.h: shard ranges, Merkle layout and
authentication-path indexing, chunk plan, MDS arms, compress_k argument rules) compiled for the CPU with
AddressSanitizer + UBSan and run (tests/cpp/test_host_logic.cpp).  GPU sanitizers are not available on the
pool, so this is where the index arithmetic of capi.hip gets its memory-safety check."""
import os
import subprocess

from conftest import ROOT


def test_host_logic_under_asan_ubsan():
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize_host.sh")], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "host logic ok" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])
