"""Randomised differential coverage inside the suite: tools/fuzz_gpu_vs_oracle.py's generators (structured carry
patterns + seeded random states and messages) through EVERY kernel family -- lane-private, lane-pair, two-row fold,
row-cooperative scan (2-1 and 4-3), one-item-per-wavefront scan, the three sponge kernels, the ragged kernel, the
segment-fed host path and the run-time-instance kernels -- against the C oracle, once, in-process, fixed seed."""
import os
import sys
import time

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_fuzz_all_kernel_families_against_the_oracle(oracle):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu_vs_oracle as fz
    log = []
    t0 = time.time()
    # 9 000 two-to-one items per field (> 8 192: the lane-private kernel at a real grid), 700 4-3 states
    failed = fz.run(n2=9000, n4=700, seed=20261004, threads=16, log=log.append)
    took = time.time() - t0
    assert not failed, "\n".join(log)
    assert len(log) == 14 + 14 + 6, log           # every (field, width) line of every section was produced
    assert took < 240, "the in-suite fuzz is meant to stay short (%.1f s; ~15 s on a 16-core box share)" % took
