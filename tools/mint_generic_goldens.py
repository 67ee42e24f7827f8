#!/usr/bin/env python3
"""Mint golden vectors for Anemoi instances wider than the shipped ones: the
reference's hard-coded `mds_layer` arms for 3..6 columns and its matrix arm have no KATs of their own
(no shipped instance reaches them).  Source of truth: oracle/anemoi_ref.py `GenericInstance`, whose
1- and 2-column cases are pinned on the reference's KATs and whose arms are tied to the matrix form in
tests/test_oracle.py.  Round constants / matrices are seeded pseudo-random field elements (the
reference defines none for these widths).  Output: tests/golden/generic.json (data only).

    python tools/mint_generic_goldens.py
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from anemoi_ref import FIELD_IDS, GenericInstance, Instance  # noqa: E402


def main():
    rng = random.Random(0xA9E301F4)
    out = []
    for field in FIELD_IDS:
        p = Instance(field, 2).p
        for cols, matrix in ((3, False), (4, False), (5, False), (6, False), (7, True)):
            rounds = 2
            C = [rng.randrange(p) for _ in range(cols * rounds)]
            D = [rng.randrange(p) for _ in range(cols * rounds)]
            M = [rng.randrange(p) for _ in range(cols * cols)] if matrix else None
            G = GenericInstance(field, cols, rounds, C, D, M)
            st = [rng.randrange(p) for _ in range(2 * cols)]
            rate = 2 * cols - 1
            msg = [rng.randrange(p) for _ in range(rate + 2)]
            out.append({
                "field": field, "num_columns": cols, "num_rounds": rounds,
                "ark_c": [str(v) for v in C], "ark_d": [str(v) for v in D],
                "mds": None if M is None else [str(v) for v in M],
                "state": [str(v) for v in st],
                "permutation": [str(v) for v in G.permutation(list(st))],
                "compress_k2": [str(v) for v in G.compress_k(st, 2)],
                "rate": rate, "message": [str(v) for v in msg], "hash_field": str(G.hash_field(msg, rate)),
            })
    with open(os.path.join(ROOT, "tests", "golden", "generic.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote %d generic-instance vectors" % len(out))


if __name__ == "__main__":
    main()
