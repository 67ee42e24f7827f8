"""Which captured NON-kernel nodes take effect on every replay of a hipGraph on this ROCm?  (a) the device-to-device copy
in front of anemoi_merkle_tree_dev's launches, (b) a bare hipMemsetAsync (issued through the HIP runtime directly:
tensor.zero_() is a fill kernel).  Each replay runs on changed inputs.  Found while putting
anemoi_hash_bytes_ragged_bucketed_dev under capture (tests/test_gpu_capture.py): its counters were zeroed by a
hipMemsetAsync, the second replay faulted ("write access to a read-only page": indices placed behind the previous replay's).
Result (profiles/r05/captured_memset_node_replays.txt): the copy node works on every replay, the memset node on the FIRST
replay only -- so no `_dev` path of the library uses hipMemsetAsync (the counters are zeroed by a kernel).
    python tools/exp_captured_nodes.py"""
import ctypes, os, sys, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import anemoi_amd as A
import orc
faulthandler.enable()
oracle = orc.Oracle()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
jub, depth = 4, 6
assert A.lib.anemoi_init(0, jub, 2) == 0
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]

d_leaves = torch.zeros((1 << depth) * 4, dtype=torch.int64, device=dev)
d_tree = torch.zeros((2 << depth) * 4, dtype=torch.int64, device=dev)
d_fill = torch.ones(65536, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=side):
    s = torch.cuda.current_stream().cuda_stream
    assert A.lib.anemoi_merkle_tree_dev(jub, d_leaves.data_ptr(), depth, d_tree.data_ptr(), s) == 0
    assert hip.hipMemsetAsync(d_fill.data_ptr(), 0, 65536 * 4, s) == 0
for i in range(4):
    leaves = rng.integers(0, 1 << 60, size=(1 << depth, 4), dtype=np.uint64)
    d_leaves.copy_(torch.from_numpy(leaves.view(np.int64).reshape(-1)).to(dev))
    d_tree.zero_(); d_fill.fill_(7)
    torch.cuda.synchronize()
    g1.replay(); torch.cuda.synchronize()
    t = d_tree.cpu().numpy().view(np.uint64).reshape(-1, 4)
    ok0 = (t[:1 << depth] == leaves).all()
    okr = (t[(2 << depth) - 2] == oracle.merkle_root(jub, leaves, depth)).all()
    print("replay %d: level 0 copied by the captured D2D copy: %s; root right: %s; bare captured memset zeroed %d of 65536 words"
          % (i, ok0, okr, int((d_fill == 0).sum().item())), flush=True)
