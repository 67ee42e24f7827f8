#!/usr/bin/env python3
"""Throughput of the run-time-instance kernels (anemoi_generic.h) next to the dedicated ones, host-pointer
entry points (PCIe copies included on both sides).  Run on the GPU box.

    python tools/bench_generic.py [field] [log2_states]
"""
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
import anemoi_amd as A  # noqa: E402


def main():
    field = sys.argv[1] if len(sys.argv) > 1 else "bls12_381"
    lg = int(sys.argv[2]) if len(sys.argv) > 2 else 18
    with open(os.path.join(ROOT, "tests", "golden", "params.json")) as f:
        P = json.load(f)[field]
    p, L = int(P["modulus"]), P["u64_limbs"]
    rng = random.Random(1)
    rows = []

    def timed(fn, reps=3):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps

    rnd = np.random.default_rng(0)
    for cols in (1, 2, 3, 4, 6, 8):
        w = 2 * cols
        n = (1 << lg) // cols
        if cols <= 2:
            inst = P["instances"]["anemoi_2_1" if cols == 1 else "anemoi_4_3"]
            rounds, C, D = inst["num_rounds"], [int(v) for v in inst["ark_c"]], [int(v) for v in inst["ark_d"]]
        else:
            rounds = 14
            C = [rng.randrange(p) for _ in range(cols * rounds)]
            D = [rng.randrange(p) for _ in range(cols * rounds)]
        ded = A.Anemoi(field, 2)
        M = None if cols <= 6 else ded.encode([rng.randrange(p) for _ in range(cols * cols)])
        g = A.GenericAnemoi(field, cols, rounds, ded.encode(C), ded.encode(D), M)
        st = ded.encode([rng.randrange(p) for _ in range(64 * w)]).reshape(64, w, L)
        st = np.ascontiguousarray(np.tile(st, (n // 64 + 1, 1, 1))[:n])
        tg = timed(lambda: g.permutation_batch(st))
        row = {"field": field, "num_columns": cols, "rounds": rounds, "states": n,
               "generic_ms": round(tg * 1e3, 2), "generic_sboxes_per_s": round(n * cols * rounds / tg / 1e6, 2)}
        if cols <= 2:
            d = A.Anemoi(field, w)
            td = timed(lambda: d.permutation_batch(st))
            assert (d.permutation_batch(st[:130]) == g.permutation_batch(st[:130])).all()
            row["dedicated_ms"] = round(td * 1e3, 2)
        rows.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
