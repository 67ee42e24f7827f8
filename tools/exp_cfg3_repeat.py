#!/usr/bin/env python3
"""Why is the FIRST config-3 launch of a process 46 % slower in KERNEL time (rocprofv3: 487 ms against 333 ms)?
Experiment: warm the same kernel with a small batch first / touch the message buffer first / neither.
    python tools/exp_cfg3_repeat.py [warm_kernel] [touch_msgs] [warm=<messages>] [api_warmup]
api_warmup: anemoi_warmup(0, bn_254, 4) first -- the library's own cure (round 5).
"""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "anemoi-rust_amd", "lib", "libanemoi_mi355x.so"))
vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
lib.anemoi_hash_bytes_dev.argtypes = [ci, ci, vp, sz, sz, vp, vp]
dev = torch.device("cuda", 0); st = torch.cuda.current_stream(); s = st.cuda_stream
rng = np.random.default_rng(7)
nmsg = 1 << 16
msgs = torch.from_numpy(rng.integers(0, 256, size=(nmsg, 10240), dtype=np.uint8)).to(dev)
dig = torch.empty(nmsg * 4, dtype=torch.int64, device=dev)
torch.cuda.synchronize()

def timed(n, ln):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    assert lib.anemoi_hash_bytes_dev(2, 4, msgs.data_ptr(), ln, n, dig.data_ptr(), s) == 0
    b.record(st)
    torch.cuda.synchronize()
    return round(a.elapsed_time(b), 1)

if "api_warmup" in sys.argv:
    import time
    lib.anemoi_warmup.argtypes = [ci, ci, ci]
    t0 = time.perf_counter()
    assert lib.anemoi_warmup(0, 2, 4) == 0
    print("anemoi_warmup(0, bn_254, 4): %.1f ms" % (1e3 * (time.perf_counter() - t0)))
if "warm_kernel" in sys.argv:
    print("warm-up: 2^15 messages of 93 bytes (the same kernel):", timed(1 << 15, 93), "ms, again", timed(1 << 15, 93))
for a in sys.argv[1:]:
    if a.startswith("warm="):          # warm=<messages>: a warm-up launch of the same kernel on that many 93-byte messages
        n = int(a[5:])
        print("warm-up: %d messages of 93 bytes:" % n, timed(n, 93), "ms")
if "touch_msgs" in sys.argv:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st); chk = int(msgs.view(torch.int64).sum().item()); b.record(st); torch.cuda.synchronize()
    print("touched the message buffer (a torch reduction): %.1f ms" % a.elapsed_time(b))
print("config 3, three calls:", [timed(nmsg, 10240) for _ in range(3)])
