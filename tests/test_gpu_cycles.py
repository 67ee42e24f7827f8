"""A slower BUILD must fail a test on ANY box (round 5's review, item 1b).  The boxes of the pool differ by up to 15 % in
the clock they hold under this load, so a throughput assertion would either be loose enough to hide a 5 % regression or
flaky; kernel time x the SLOWEST sampled XCD clock (millions of shader cycles) is box-independent to +-1 % (DESIGN.md
section 5, profiles/r05/clock_sampler_across_boxes.txt).  Budgets = the committed measurement of the round's final sources
(profiles/r06/cycles_budget.json) + 2 %."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

@pytest.fixture(scope="module")
def A():
    import anemoi_amd
    assert anemoi_amd.device_count() >= 1
    return anemoi_amd


# millions of cycles of the slowest XCD per launch.  Measured on the final sources (profiles/r06/cycles_budget.json, the fastest
# of 5 repetitions): headline 227.7, config 3 782.6, config 5's widest level 77.4; across the round's boxes and sessions
# 227.0-229.6 / 783-787 / 77.4-77.6.  Budget = the top of that range + 2 % (headline: + 1.5 %, the review's 233).
# The BENCH LINE's figure is the MEAN of its timed steps x the mean of the clock over them.  Until bench.py made everything
# it needs BEFORE its warm-up, a gap of a few milliseconds sat in front of the first timed step, which then ran 3-6 % slower
# (the chip re-entering full load: profiles/r06/first_step_clock.txt) and a three-step line read 231.8-232.8; without the gap
# it reads 227.2-228.0 like the in-process figure, and has the same budget.
BUDGET_MCYCLES = {"headline": 233.0, "cfg3": 803.0, "cfg5_top": 79.2, "headline_bench_line_steps3": 233.0}
VALU_INSTR_PER_ITEM = 3629944        # SQ_INSTS_VALU per compression of k_jive<bls12_381,2,2> (profiles/rNN/pmc_k_jive.json)


def test_headline_cycles_and_instruction_count_from_bench_line():
    """bench.py --steps 3 as the driver runs it (a child process): the line's own box-independent figures."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    alu = line["alu"]
    print("bench line: %.3f M/s, kernel %.3f ms, %.2f Mcycles at the slowest XCD's %.3f GHz" % (
        line["value"] / 1e6, line["roofline"]["kernel_ms"], alu["kernel_Mcycles_slowest_xcd"], alu["clock_GHz_measured_slowest_xcd"]))
    assert line["verified"]["sha256_of_all_outputs"] is True
    assert alu["clock_sampler_groups"] >= 8
    assert 150.0 < alu["kernel_Mcycles_slowest_xcd"] <= BUDGET_MCYCLES["headline_bench_line_steps3"], alu["kernel_Mcycles_slowest_xcd"]
    # the committed counter profile must be of THESE sources (else the line carries null), and say what it said
    assert line["roofline"]["traffic_stale"] is False, "profiles/rNN/pmc_k_jive.json was not taken from the kernel sources being run"
    assert alu["valu_instr_per_item"] == VALU_INSTR_PER_ITEM, alu["valu_instr_per_item"]
    assert 0.9 < alu["valu_issue_frac"] <= 1.005 and alu["valu_issue_inconsistent"] is False
    assert 1.0 <= line["roofline"]["traffic"] / line["roofline"]["algorithmic_bytes_per_launch"] < 1.05


@pytest.mark.parametrize("name", ["cfg3", "cfg5_top", "headline"])
def test_kernel_cycles_budget(A, name):
    """the same figure for config 3's k_sponge_pair, config 5's 2^20-node level and (in process) the headline"""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import measure_cycles
    r = measure_cycles.measure(A, torch.device("cuda", 0), names=(name,), reps=3, best=True)["kernels"][name]
    print("%s: %.3f ms x %.3f GHz = %.2f Mcycles (budget %.1f)" % (name, r["ms"], r["clock_GHz_slowest_xcd"], r["Mcycles_slowest_xcd"],
                                                                   BUDGET_MCYCLES[name]))
    assert 0.5 * BUDGET_MCYCLES[name] < r["Mcycles_slowest_xcd"] <= BUDGET_MCYCLES[name], r
