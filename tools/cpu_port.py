#!/usr/bin/env python3
"""The CPU baseline beside the GPU numbers of tools/bench_configs.py and tools/bench_reference_workloads.py.

What is timed is the C oracle (oracle/anemoi_oracle.c: u64 CIOS Montgomery, the arkworks algorithm restated, built
-O3 -march=native with the image's clang) -- kind "port".  The reference's own Rust binary cannot be built on the
boxes (no rustc / cargo, arkworks not vendored); `reference_toolchain()` reports the probe.  The oracle is test
infrastructure: these tools time it AFTER and OUTSIDE every GPU-timed region, and nothing under anemoi-rust_amd/
imports it."""
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np


def usable_threads():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(cores, int(os.environ.get("ANEMOI_CPU_THREADS", "16"))))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def reference_toolchain():
    return {"cargo": shutil.which("cargo"), "rustc": shutil.which("rustc")}


class Port:
    def __init__(self):
        import orc
        path = None
        try:
            path = orc.build(native=True, out=os.path.join("/tmp", "liboracle_native_%d.so" % os.getpid()))
        except Exception:
            path = None
        self.orc = orc
        self.oracle = orc.Oracle(path)
        self.threads = usable_threads()
        self.rng = np.random.default_rng(11)

    def describe(self):
        return "C port (oracle/anemoi_oracle.c, %s) on %s: %d threads / 1 thread; reference Rust toolchain: %s" % (
            self.orc.compiler_description(), cpu_model(), self.threads, reference_toolchain())

    def _states(self, n, width, limbs):
        return self.rng.integers(0, 1 << 60, size=(n, width, limbs), dtype=np.uint64)   # limbs < 2^60 => element < p

    def compress_us(self, fid, width, limbs, k=2, budget_s=2.0):
        """(microseconds per compression on one thread, compressions/s on all threads)"""
        st = self._states(64, width, limbs)
        t0 = time.perf_counter()
        self.oracle.compress_batch(fid, width, st, k=k, threads=1)
        one = (time.perf_counter() - t0) / 64
        n1 = max(64, int(budget_s / 3 / one))
        st = self._states(n1, width, limbs)
        t0 = time.perf_counter()
        self.oracle.compress_batch(fid, width, st, k=k, threads=1)
        one = (time.perf_counter() - t0) / n1
        nt = max(64 * self.threads, int(budget_s / one * self.threads))
        st = self._states(nt, width, limbs)
        t0 = time.perf_counter()
        self.oracle.compress_batch(fid, width, st, k=k, threads=self.threads)
        return one * 1e6, nt / (time.perf_counter() - t0)

    def hash_us(self, fid, width, msg_len, budget_s=2.0):
        """(microseconds per message on one thread, messages/s on all threads)"""
        m = self.rng.integers(0, 256, size=(2, msg_len), dtype=np.uint8)
        t0 = time.perf_counter()
        self.oracle.hash_bytes_batch(fid, width, m, threads=1)
        one = (time.perf_counter() - t0) / 2
        nt = max(self.threads, int(budget_s / one * self.threads))
        m = self.rng.integers(0, 256, size=(nt, msg_len), dtype=np.uint8)
        t0 = time.perf_counter()
        self.oracle.hash_bytes_batch(fid, width, m, threads=self.threads)
        return one * 1e6, nt / (time.perf_counter() - t0)

    def merkle_ms(self, fid, limbs, depth):
        """milliseconds for a depth-`depth` tree, level by level, every level on all threads"""
        lv = self._states(1 << depth, 1, limbs).reshape(1 << depth, limbs)
        t0 = time.perf_counter()
        n = 1 << depth
        while n > 1:
            lv = self.oracle.compress_batch(fid, 2, lv.reshape(n // 2, 2, limbs), threads=min(self.threads, max(1, n // 2))).reshape(n // 2, limbs)
            n //= 2
        return (time.perf_counter() - t0) * 1e3
