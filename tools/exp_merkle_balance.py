#!/usr/bin/env python3
"""Does the do-nothing launch in front of underfilled launches (option balance_underfilled) cost or gain a Merkle tree anything?
Device-resident trees, timed back to back, the option on / off alternating in one process.
    python tools/exp_merkle_balance.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
import numpy as np
import torch
import anemoi_amd as A
from anemoi_amd import synth


def main():
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()
    for field, limbs, depth in (("jubjub", 4, 21), ("bls12_381", 6, 18), ("jubjub", 4, 17), ("bls12_381", 6, 16)):
        fid = A.field_id(field)
        n = 1 << depth
        leaves = synth.states(field, 2, 99, 0, n // 2).reshape(n, limbs)
        d_leaves = torch.from_numpy(leaves.view(np.int64).reshape(-1)).to(dev)
        d_scr = torch.empty(n * limbs, dtype=torch.int64, device=dev)
        d_root = torch.zeros(limbs, dtype=torch.int64, device=dev)
        assert A.lib.anemoi_init(0, fid, 2) == 0

        def tree():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            assert A.lib.anemoi_merkle_root_dev(fid, d_leaves.data_ptr(), depth, d_scr.data_ptr(), d_root.data_ptr(), st.cuda_stream) == 0
            b.record(st)
            torch.cuda.synchronize()
            return a.elapsed_time(b)

        tree(), tree()
        res = {0: [], 1: []}
        roots = {}
        for rnd in range(6):
            for on in (1, 0):
                with A.options(balance_underfilled=on):
                    res[on].append(tree())
                    roots[on] = d_root.cpu().numpy().tobytes()
        assert roots[0] == roots[1]
        med = lambda v: sorted(v)[len(v) // 2]
        print("%-10s depth %2d: balanced %.3f ms (min %.3f) | option off %.3f ms (min %.3f) | difference %+.3f ms"
              % (field, depth, med(res[1]), min(res[1]), med(res[0]), min(res[0]), med(res[1]) - med(res[0])))


if __name__ == "__main__":
    main()
