#!/usr/bin/env python3
"""Merkle root timing (device-resident leaves) for a field/depth; run under different ANEMOI_COOP_MAX
values to see what the wave-cooperative latency kernel buys on the top levels of the tree."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
field, limbs, depth = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lib = ctypes.CDLL(os.path.join(ROOT, "anemoi-rust_amd", "lib", "libanemoi_mi355x.so"))
vp = ctypes.c_void_p
lib.anemoi_merkle_root_dev.argtypes = [ctypes.c_int, vp, ctypes.c_uint, vp, vp, vp]
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
leaves = torch.from_numpy(rng.integers(0, 1 << 60, size=(1 << depth, limbs), dtype=np.uint64).view(np.int64).reshape(-1)).to(dev)
scratch = torch.empty((1 << depth) * limbs, dtype=torch.int64, device=dev)
root = torch.empty(limbs, dtype=torch.int64, device=dev)
s = torch.cuda.current_stream()
def run():
    assert lib.anemoi_merkle_root_dev(field, leaves.data_ptr(), depth, scratch.data_ptr(), root.data_ptr(), s.cuda_stream) == 0
run(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s); run(); b.record(s); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print("field %d depth %d COOP_MAX=%s: %.2f ms  root[0]=%x" % (field, depth, os.environ.get("ANEMOI_COOP_MAX", "default"), sorted(ts)[1], root[0].item() & 0xffffffffffffffff))
