#!/usr/bin/env python3
"""Mint golden vectors for the cases the reference's own tests leave unpinned:
hash(bytes) with a partial last chunk, empty input, the 10 240-byte bench message, compress on
random states, small Merkle roots.  Source of truth: the Python big-int restatement
oracle/anemoi_ref.py, itself pinned against all 420 reference KATs (tests/test_oracle.py) and
cross-checked against the C oracle.  Output: tests/golden/extra.json (data only).

    python tools/mint_goldens.py
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from anemoi_ref import FIELD_IDS, Instance  # noqa: E402


def main():
    rng = random.Random(0xA9E301FF)
    out = {"hash_bytes": [], "compress": [], "merkle": []}
    for field in FIELD_IDS:
        for width in (2, 4):
            I = Instance(field, width)
            for ln in (0, 1, I.chunk - 1, I.chunk + 1, 2 * I.chunk + 7):
                msg = bytes(rng.randrange(256) for _ in range(ln))
                out["hash_bytes"].append({"field": field, "width": width, "msg_hex": msg.hex(), "digest": str(I.hash(msg))})
            st = [rng.randrange(I.p) for _ in range(width)]
            out["compress"].append({"field": field, "width": width, "in": [str(v) for v in st],
                                    "out": [str(v) for v in I.compress(st)]})
    # the bench/config-3 message shape: 10 240 bytes (10240 % 31 = 10, % 47 = 41 -> padding branch)
    for field, width in (("bn_254", 4), ("bls12_381", 2), ("vesta", 4)):
        I = Instance(field, width)
        msg = bytes(rng.randrange(256) for _ in range(10240))
        out["hash_bytes"].append({"field": field, "width": width, "msg_hex": msg.hex(), "digest": str(I.hash(msg))})
    for field, depth in (("jubjub", 4), ("bls12_381", 3), ("pallas", 5)):
        I = Instance(field, 2)
        leaves = [rng.randrange(I.p) for _ in range(1 << depth)]
        out["merkle"].append({"field": field, "depth": depth, "leaves": [str(v) for v in leaves],
                              "root": str(I.merkle_root(leaves))})
    with open(os.path.join(ROOT, "tests", "golden", "extra.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
