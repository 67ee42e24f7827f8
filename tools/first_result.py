#!/usr/bin/env python3
"""What a fresh process pays before its first result: dlopen of the library (the gfx950 code objects are registered
then), the first anemoi_jive_compress_batch on host pointers (HIP context, constant tables, kernel load) and the same
call again.  One line per library, each in its own child process:
    python tools/first_result.py [lib.so ...]          (default: the product library)
"""
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(path):
    import numpy as np
    t0 = time.perf_counter()
    lib = ctypes.CDLL(path)
    t1 = time.perf_counter()
    fn = lib.anemoi_jive_compress_batch
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    n = 1 << 16
    st = np.random.default_rng(1).integers(0, 1 << 60, size=(n, 2, 6), dtype=np.uint64)
    out = np.zeros((n, 6), dtype=np.uint64)
    ts = []
    for _ in range(3):
        a = time.perf_counter()
        assert fn(0, 2, st.ctypes.data, out.ctypes.data, n, 0) == 0
        ts.append(time.perf_counter() - a)
    print("%-28s %8.1f KiB  dlopen %7.1f ms | first compress_batch of 2^16 %8.1f ms | second %6.1f | third %6.1f | checksum %016x" % (
        os.path.basename(path), os.path.getsize(path) / 1024, 1e3 * (t1 - t0), 1e3 * ts[0], 1e3 * ts[1], 1e3 * ts[2],
        int(np.bitwise_xor.reduce(out.reshape(-1)))))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        libs = sys.argv[1:] or [os.path.join(ROOT, "anemoi-rust_amd", "lib", "libanemoi_mi355x.so")]
        for rep in range(2):
            for path in libs:
                subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", os.path.abspath(path)])
