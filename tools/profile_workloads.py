#!/usr/bin/env python3
"""The workloads whose kernels are NOT the headline, run a few times each so that rocprofv3 can be wrapped around
them (tools/collect_config_profiles.sh): config 3 (BN-254 Anemoi-4-3 sponge, 2^16 messages of 10 240 bytes:
k_sponge_pair<2, true>), config 5 (one GPU's depth-21 Jubjub subtree: k_jive<4, 2, 2> level by level, then the
wave-cooperative kernel), and the flat batches their rates are compared with (Jubjub 2-1 and BN-254 4-3 Jive, 2^20).
Inputs are resident in HBM; nothing here is timed -- the profiler does that.

    python tools/profile_workloads.py [cfg3] [cfg5] [flat] [--reps 2]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
# the summariser picks a config's dispatch by its grid and prices its bytes from these
from summarize_config_profiles import CFG3_MESSAGES, CFG3_MSG_LEN, WARM_MESSAGES, WARM_MSG_LEN


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
    want = set(args) or {"cfg3", "cfg5", "flat"}
    lib = ctypes.CDLL(os.path.join(ROOT, "anemoi-rust_amd", "lib", "libanemoi_mi355x.so"))
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    lib.anemoi_jive_compress_k_dev.argtypes = [ci, ci, ci, vp, vp, sz, vp]
    lib.anemoi_hash_bytes_dev.argtypes = [ci, ci, vp, sz, sz, vp, vp]
    lib.anemoi_merkle_root_dev.argtypes = [ci, vp, ctypes.c_uint, vp, vp, vp]
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(7)

    def states(n, width, limbs):
        h = rng.integers(0, 1 << 60, size=(n, width, limbs), dtype=np.uint64)
        return torch.from_numpy(h.view(np.int64).reshape(-1)).to(dev)

    if "flat" in want:
        for field, width, limbs, n in ((4, 2, 4, 1 << 20), (2, 4, 4, 1 << 20)):
            d_in, d_out = states(n, width, limbs), torch.empty(n * limbs * (width // 2), dtype=torch.int64, device=dev)
            for _ in range(reps + 1):
                assert lib.anemoi_jive_compress_k_dev(field, width, 2, d_in.data_ptr(), d_out.data_ptr(), n, s) == 0
            torch.cuda.synchronize()
    if "cfg3" in want:
        nmsg = CFG3_MESSAGES
        msgs = torch.from_numpy(rng.integers(0, 256, size=(nmsg, CFG3_MSG_LEN), dtype=np.uint8)).to(dev)
        dig = torch.empty(nmsg * 4, dtype=torch.int64, device=dev)
        # The FIRST launch of this kernel in a process takes 487-491 ms of kernel time instead of 333 (rocprofv3 trace;
        # tools/exp_cfg3_repeat.py, profiles/r04/first_launch_after_idle.txt).  Other kernels launched before do not change
        # that (a full-chip Jive batch, a torch reduction over the buffer); a small launch of the SAME kernel first does (it
        # is itself 4.2 instead of 2.0 ms).  x 1.47 is what three instead of two long-running wavefronts per SIMD cost, i.e.
        # it looks like the first dispatch of a kernel placing its workgroups on part of the chip; the cause is not
        # established.  Two small launches first, so that the timed ones measure the steady state.
        for _ in range(2):
            assert lib.anemoi_hash_bytes_dev(2, 4, msgs.data_ptr(), WARM_MSG_LEN, WARM_MESSAGES, dig.data_ptr(), s) == 0
        for _ in range(reps):
            assert lib.anemoi_hash_bytes_dev(2, 4, msgs.data_ptr(), CFG3_MSG_LEN, nmsg, dig.data_ptr(), s) == 0
        torch.cuda.synchronize()
    if "cfg5" in want:
        depth = 21
        leaves = states(1 << depth, 1, 4)
        scratch = torch.empty((1 << depth) * 4, dtype=torch.int64, device=dev)
        root = torch.empty(4, dtype=torch.int64, device=dev)
        for _ in range(reps):
            assert lib.anemoi_merkle_root_dev(4, leaves.data_ptr(), depth, scratch.data_ptr(), root.data_ptr(), s) == 0
        torch.cuda.synchronize()
    print("done:", sorted(want))


if __name__ == "__main__":
    main()
