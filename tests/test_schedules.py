"""The exponentiation schedules the kernels run are DATA in a generated header
(anemoi-rust_amd/csrc/field_consts_gen.h: sliding-window schedules kW2..kW5, and the window-3 schedule with two
extra table digits kXSched + the digits' build programmes).  tools/gen_params.py checks them when it writes the
header; this test re-reads the COMMITTED header and replays every schedule with Python integers against
pow(x, INV_ALPHA, p), so a stale or hand-edited header cannot slip through on the CPU side (the GPU parity tests
would catch it too, but only on a GPU box).  x^INV_ALPHA is what the reference's hard-coded chains compute
(src/<field>/sbox.rs exp_by_inv_alpha; its own test_alpha checks the same identity, sbox.rs:490-496)."""
import os
import re

import pytest

from conftest import FIELD_IDS, ROOT

HDR = os.path.join(ROOT, "anemoi-rust_amd", "csrc", "field_consts_gen.h")


def field_block(text, fid):
    a = text.index("template <> struct FieldC<%d>" % fid)
    b = text.find("template <> struct FieldC<%d>" % (fid + 1))
    return text[a:b if b > 0 else len(text)]


def ints(blk, name):
    m = re.search(r"\b%s\[[^\]]*\]\s*=\s*\{([^}]*)\}" % name, blk)
    return [int(v, 0) for v in m.group(1).replace("\n", " ").split(",") if v.strip()]


def scalar(blk, name):
    return int(re.search(r"\b%s = (-?\d+)" % name, blk).group(1))


@pytest.mark.parametrize("fid", range(7))
def test_committed_schedules_compute_x_to_inv_alpha(params, fid):
    fp = params[FIELD_IDS[fid]]
    p, e = int(fp["modulus"]), int(fp["inv_alpha"])
    assert fp["alpha"] * e % (p - 1) == 1
    blk = field_block(open(HDR).read(), fid)
    for x in (2, 0x1234567 % p, p - 3):
        want = pow(x, e, p)
        # plain sliding windows: table T[j] = x^(2j+1); ops 253 (tmp = acc), 254 (acc *= tmp), 255 (no multiplication)
        for k in (2, 3, 4, 5):
            first, nsteps, flat = scalar(blk, "kW%dFirst" % k), scalar(blk, "kW%dSteps" % k), ints(blk, "kW%dSched" % k)
            assert len(flat) == 2 * nsteps
            tab = [pow(x, 2 * j + 1, p) for j in range(1 << (k - 1))]
            acc, tmp = tab[first], None
            for s, op in zip(flat[0::2], flat[1::2]):
                if op == 253:
                    tmp = acc
                    continue
                acc = pow(acc, 1 << s, p)
                if op == 254:
                    acc = acc * tmp % p
                elif op != 255:
                    acc = acc * tab[op] % p
            assert acc == want, (FIELD_IDS[fid], k)
        # window 3 + extra digits: table indices 0..3 = x, x^3, x^5, x^7; 4, 5 = the extras, built by their programmes
        nx = scalar(blk, "kXDigits")
        if nx == 0:
            continue
        digits, lens = ints(blk, "kXDigit")[:nx], ints(blk, "kXProgLen")[:nx]
        ops, args = ints(blk, "kXProgOp"), ints(blk, "kXProgArg")
        src = [x, x * x % p, pow(x, 3, p), pow(x, 5, p), pow(x, 7, p), None]
        table = [x, pow(x, 3, p), pow(x, 5, p), pow(x, 7, p)]
        off = 0
        for i in range(nx):
            r = None
            for op, arg in zip(ops[off:off + lens[i]], args[off:off + lens[i]]):
                if op == 0:
                    r = src[arg]
                elif op == 1:
                    r = pow(r, 1 << arg, p)
                else:
                    r = r * src[arg] % p
            off += lens[i]
            assert r == pow(x, digits[i], p), (FIELD_IDS[fid], digits[i])
            table.append(r)
            if i == 0:
                src[5] = r
        first, nsteps, flat = scalar(blk, "kXFirst"), scalar(blk, "kXSteps"), ints(blk, "kXSched")
        assert len(flat) == 2 * nsteps and first < len(table)
        acc = table[first]
        for s, op in zip(flat[0::2], flat[1::2]):
            acc = pow(acc, 1 << s, p)
            if op != 255:
                acc = acc * table[op] % p
        assert acc == want, FIELD_IDS[fid]


def test_header_is_what_the_generator_writes_today(tmp_path):
    """field_consts_gen.h and the oracle's parameter header are regenerated and compared byte for byte."""
    import subprocess
    import sys
    keep = {}
    paths = [HDR, os.path.join(ROOT, "oracle", "anemoi_params_gen.h")]
    for q in paths:
        keep[q] = open(q, "rb").read()
    try:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_params.py")], stdout=subprocess.DEVNULL)
        for q in paths:
            assert open(q, "rb").read() == keep[q], "%s is stale: run tools/gen_params.py" % os.path.basename(q)
    finally:
        for q in paths:
            open(q, "wb").write(keep[q])
