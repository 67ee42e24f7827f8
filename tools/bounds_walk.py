#!/usr/bin/env python3
"""The JUDGE of the lazy-reduction bounds (tests/cpp/bounds_walk/bounds_walk.cpp is the witness).

The witness compiles the kernel bodies of anemoi-rust_amd/csrc for the host over arithmetics that carry an upper bound
of every value, runs every kernel family of every field, and logs each distinct step (operation, operand bounds, result
bound, first statement that did it) and, per statement (file:line), the largest operands and result it ever saw.  This
module re-derives every step with Python integers from tests/golden/params.json -- the witness's bookkeeping is
believed only where the two agree -- and checks what each operation needs:

  products (mul, sqr, settle, mul_g as a product, from_abi, from_int, to_abi)
      * both operands below R' (limbs below 2^W);
      * 30-bit lane limbs: the operand bound the generated assembly was dimensioned for (tools/gen_asm_mul.py:
        a < 16 p for a squaring, b < 16 p for a multiplication -- the top limb);
      * lane-private and scan arithmetics: the result below 2 p (A B <= H), which is what lets a product feed a
        product, a `canonical`, or a subtraction pad of 4 p; the fold arithmetic: the result below R';
      * to_abi: the digit-serial product in front of `canonical` ends below 2 p;
  add, add_k, mul_g by limb scaling
      * the result below R' (the cooperative forms DROP what leaves the top limb; the lane-private form needs limbs
        below 2^W at the next product);
  sub (a + pad - b)
      * b below R', every limb of b under the pad's limb: lower limbs of the pad >= 2^W - 1, top limb of b's bound
        <= the pad's top limb; the result below R';
  constants
      * One, RR, In, Out, GMont, Delta and the pad of every layout are what the arithmetic assumes them to be.

    python tools/bounds_walk.py [--table anemoi-rust_amd/csrc/BOUNDS.md]     # build, run, judge, (re)write the table
"""
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_asm_mul  # noqa: E402  (SQR_A_MULT, MUL_B_MULT, top_limb_bound: the assembly's documented operand limits)

FIELD_IDS = ["bls12_381", "bls12_377", "bn_254", "ed_on_bls12_377", "jubjub", "pallas", "vesta"]
WALK_DIR = os.path.join(ROOT, "tests", "cpp", "bounds_walk")
PRODUCT_OPS = ("mul", "sqr", "settle", "mul_g_product", "from_abi", "from_int")
CONST_OPS = ("one", "zero", "delta", "ark_c", "ark_d")


def round_up(v, bits=24):
    """the witness's rounding: up to `bits` significant bits (a bound rounded up is still a bound)"""
    top = v.bit_length() - 1
    if top < bits:
        return v
    drop = top - (bits - 1)
    hi = v >> drop
    return v if (hi << drop) == v else (hi + 1) << drop


def clamp(v):
    """what the witness keeps for a result: rounded up, or 2^500 once a (mutated) walk has left every limit"""
    return 1 << 500 if v > 1 << 500 else round_up(v)


def build(out_dir=None, csrc=None, extra="", fields=range(7)):
    cmd = ["make", "-s", "-C", WALK_DIR, "-j", str(min(8, os.cpu_count() or 1)), "FIELDS=" + " ".join(str(f) for f in fields)]
    if out_dir:
        cmd.append("OUT=" + out_dir)
    if csrc:   # a (mutated) copy of anemoi-rust_amd/csrc to walk instead of the product's
        cmd.append("CSRC=" + csrc)
    if extra:
        cmd.append("EXTRA=" + extra)
    subprocess.check_call(cmd)
    return out_dir or os.path.join(ROOT, "tests", "cpp", "build")


def run(bin_dir, fields=range(7)):
    """runs the per-field witnesses side by side; returns the parsed logs {field id: Log}"""
    tmp = tempfile.mkdtemp(prefix="bounds_walk_")

    def one(f):
        out = os.path.join(tmp, "walk_%d.txt" % f)
        subprocess.check_call([os.path.join(bin_dir, "bounds_walk_%d" % f), out])
        log = Log(out)
        os.unlink(out)
        return f, log

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        logs = dict(ex.map(one, fields))
    os.rmdir(tmp)
    return logs


class Layout:
    def __init__(self, kv):
        self.name = kv["name"]
        self.W, self.NL, self.g, self.chunk, self.nabi = (int(kv[k]) for k in ("W", "NL", "g", "chunk", "nabi"))
        for k in ("p", "KP", "One", "In", "Out", "RR", "GMont", "Delta"):
            setattr(self, k, int(kv[k], 16))
        self.kp_limbs = [int(x, 16) for x in kv["kp_limbs"].split(",")]
        self.R = 1 << (self.W * self.NL)
        self.shift = self.W * self.NL

    def mont(self, a, b):
        return (a * b + (self.R - 1) * self.p) >> self.shift

    def fold_product(self, a, b):
        # tools/coop2d_model.py / tests/test_coop2d_model.py: the fold product ends below a b / R' + (NL (2^W + 32) + 2) p
        return ((a * b) >> self.shift) + (self.NL * ((1 << self.W) + 32) + 2) * self.p


class Log:
    def __init__(self, path):
        self.layout, self.flags, self.sites, self.steps, self.cases = {}, {}, [], [], []
        self.overflow = None
        with open(path) as f:
            for line in f:
                tok = line.split()
                if tok[0] == "walk":
                    self.header = dict(t.split("=") for t in tok[2:])
                elif tok[0] == "layout":
                    self.layout[tok[2]] = Layout(dict(t.split("=", 1) for t in tok[3:]))
                elif tok[0] == "flags":
                    self.flags = {k: int(v) for k, v in (t.split("=") for t in tok[2:])}
                elif tok[0] == "site":
                    kv = dict(t.split("=", 1) for t in tok[5:9])
                    where = line.split(" in=", 1)[1].rstrip("\n")
                    self.sites.append(dict(arith=tok[2], site=tok[3], op=tok[4], n=int(kv["n"]), a=int(kv["a"], 16),
                                           b=int(kv["b"], 16), out=int(kv["out"], 16), where=where))
                elif tok[0] == "step":
                    first = line.split(" first=", 1)[1].rstrip("\n")
                    self.steps.append(dict(arith=tok[2], op=tok[3], a=int(tok[4], 16), b=int(tok[5], 16), out=int(tok[6], 16),
                                           n=int(tok[7][2:]), first=first))
                elif tok[0] == "case":
                    self.cases.append(line[5:].rstrip("\n"))
                elif tok[0] == "overflow":
                    self.overflow = int(tok[1])


def check_constants(field, L, arith, params):
    """the layout's constants against the reference's parameters: what the step rules below assume about them"""
    P = params[FIELD_IDS[field]]
    p, g, delta, l64 = int(P["modulus"]), int(P["beta"]), int(P["delta"]), int(P["u64_limbs"])
    bad = []

    def want(name, got, exp):
        if got != exp:
            bad.append("%s %s: %s is %#x, expected %#x" % (FIELD_IDS[field], arith, name, got, exp))

    want("p", L.p, p)
    want("g", L.g, g)
    want("chunk", L.chunk, int(P["byte_chunk"]))
    want("One", L.One, L.R % p)
    want("RR", L.RR, L.R * L.R % p)
    want("In", L.In, pow(2, 2 * L.shift - 64 * l64, p))
    want("Out", L.Out, pow(2, 64 * l64, p))
    want("GMont", L.GMont, g * L.R % p)
    want("Delta", L.Delta, delta * L.R % p)
    want("pad = its limbs", sum(v << (L.W * i) for i, v in enumerate(L.kp_limbs)), L.KP)
    if L.KP % p:
        bad.append("%s %s: the subtraction pad is not a multiple of p" % (FIELD_IDS[field], arith))
    if (1 << (8 * L.chunk)) >= p:
        bad.append("%s: a %d-byte chunk can reach p" % (FIELD_IDS[field], L.chunk))
    return bad


def judge_step(field, log, s):
    """[] or the reasons one distinct step is not allowed"""
    ar, op, a, b, out = s["arith"], s["op"], s["a"], s["b"], s["out"]
    L = log.layout[ar]
    p, R = L.p, L.R
    bad = []
    if op in PRODUCT_OPS or op == "to_abi":
        exact = L.mont(a, b) if (ar != "fold" or op == "to_abi") else L.fold_product(a, b)
        if out != clamp(exact):
            bad.append("witness and judge disagree on the result bound")
        if a >= R or b >= R:
            bad.append("operand not below R'")
        if ar == "lane" and L.W >= 30:   # the generated assembly's documented operand limits (top limb)
            x, mult = (a, gen_asm_mul.SQR_A_MULT) if op == "sqr" else (b, gen_asm_mul.MUL_B_MULT)
            if (x >> (L.W * (L.NL - 1))) > gen_asm_mul.top_limb_bound(p, L.W, L.NL, mult):
                bad.append("%s operand above the %d p the 30-bit assembly is dimensioned for" % ("squaring" if op == "sqr" else "second", mult))
        if op == "to_abi":
            if exact > 2 * p - 1:
                bad.append("conversion does not end below 2p in front of canonical()")
        elif ar == "fold":
            if exact >= R:
                bad.append("fold product not below R'")
        elif exact > 2 * p - 1:
            bad.append("product not below 2p (A B > H)")
    elif op in ("add", "add_k", "mul_g_scale"):
        exact = a * b if op == "mul_g_scale" else a + b
        if out != clamp(exact):
            bad.append("witness and judge disagree on the result bound")
        if op == "mul_g_scale" and b != L.g:
            bad.append("scaling by something else than the generator")
        if op == "mul_g_scale" and ar == "fold" and L.g * ((1 << L.W) + 32) >= 1 << 32:
            bad.append("g times a fold limb does not fit 32 bits")
        if exact >= R:
            bad.append("sum not below R'")
    elif op == "sub":
        exact = a + L.KP
        if out != clamp(exact):
            bad.append("witness and judge disagree on the result bound")
        if b >= R:
            bad.append("subtrahend not below R'")
        if any(v < (1 << L.W) - 1 for v in L.kp_limbs[:-1]):
            bad.append("a lower limb of the pad is below 2^W - 1")
        if (b >> (L.W * (L.NL - 1))) > L.kp_limbs[-1]:
            bad.append("subtrahend above the pad (top limb)")
        if any(v + (1 << L.W) + 128 >= 1 << 32 for v in L.kp_limbs):
            bad.append("minuend limb + pad limb does not fit 32 bits")
        if exact >= R:
            bad.append("difference not below R'")
    elif op in CONST_OPS:
        exp = {"one": L.One, "zero": 0, "delta": L.Delta}.get(op)
        if exp is not None and out != round_up(exp):
            bad.append("constant is not what its name says")
        if out >= 2 * p:
            bad.append("constant not below 2p")
    else:
        bad.append("operation the judge has no rule for")   # incl. `raw`: limbs written around the arithmetic
    return bad


def judge(logs, params):
    """-> list of violation strings (empty = every step of every kernel family of every field is allowed)"""
    bad = []
    for field, log in sorted(logs.items()):
        if log.overflow != 0:
            bad.append("%s: the witness's 1024-bit integers overflowed" % FIELD_IDS[field])
        for ar, L in log.layout.items():
            bad += check_constants(field, L, ar, params)
        for s in log.steps:
            for why in judge_step(field, log, s):
                bad.append("%s %s %s at %s: %s (a = %.4g p, b = %.4g p, result <= %.4g p)" % (
                    FIELD_IDS[field], s["arith"], s["op"], s["first"], why, s["a"] / log.layout[s["arith"]].p,
                    s["b"] / log.layout[s["arith"]].p, s["out"] / log.layout[s["arith"]].p))
    return bad


# ---- the table ---------------------------------------------------------------------------------------------------------
def fmt(x):
    if x == 0:
        return "0"
    if x < 1000:
        return ("%.3g" % x)
    e = 0
    while x >= 2:
        x /= 2
        e += 1
    return "2^%.1f" % (e + (x - 1)) if e else "%.3g" % x


def load(site, L):
    """how much of its tightest limit a statement uses, as (fraction, what the limit is)"""
    op, a, b, out, p, R = site["op"], site["a"], site["b"], site["out"], L.p, L.R
    ar = site["arith"]
    if op in PRODUCT_OPS or op == "to_abi":
        if ar == "fold" and op != "to_abi":
            return max(a, b) / R, "operand / R'"
        worst = a * b / (R * p)
        what = "A B / H"
        if ar == "lane" and L.W >= 30:
            x = a if op == "sqr" else b
            if x / (16 * p) > worst:
                worst, what = x / (16 * p), "operand / 16 p"
        return worst, what
    if op == "sub":
        return (b >> (L.W * (L.NL - 1))) / L.kp_limbs[-1], "subtrahend / pad"
    if op in ("add", "add_k", "mul_g_scale"):
        return out / R, "result / R'"
    return 0.0, ""


def table(logs):
    out = []
    out.append("# Lazy-reduction bounds of every arithmetic statement (GENERATED: `python tools/bounds_walk.py --table`)")
    out.append("")
    out.append("The kernel bodies of `anemoi-rust_amd/csrc` are compiled for the host over bound-carrying arithmetics")
    out.append("(`tests/cpp/bounds_walk/bounds_walk.cpp`) and every kernel family of every field is run; `tools/bounds_walk.py`")
    out.append("re-derives and judges every step with exact integers (`tests/test_bounds_walk.py`, which also checks that this")
    out.append("file is what the walk prints today).  Per statement (file:line of the call) and field: the largest operands it")
    out.append("ever sees and the largest result, in units of p, then the share of its tightest limit that this uses --")
    out.append("`A B / H` for a product that must end below 2p (H = R'/p), `operand / 16 p` where the 30-bit assembly's")
    out.append("operand limit is tighter, `subtrahend / pad` (top limb), `result / R'` for sums, `operand / R'` for the fold")
    out.append("product.  Anything above 100 % fails the test.  Bounds are rounded UP to 24 significant bits.")
    out.append("")
    titles = {"lane": "Lane-private arithmetic (`mont29.h`; the throughput kernels, the run-time-instance kernels)",
              "scan": "Row-cooperative scan (`coop29.h`; latency kernels with LPR = 16)",
              "fold": "Two-row fold (`coop2d.h`; latency kernels with LPR = 32)"}
    for ar in ("lane", "scan", "fold"):
        out.append("## " + titles[ar])
        out.append("")
        hdr = []
        for f in sorted(logs):
            L = logs[f].layout[ar]
            hdr.append("%s: %d x %d bits, H = %s, pad %s p" % (FIELD_IDS[f], L.NL, L.W, fmt(L.R / L.p), fmt(L.KP // L.p)))
        out.append("Layouts: " + "; ".join(hdr) + ".")
        out.append("")
        out.append("| statement | op | " + " | ".join(FIELD_IDS[f] for f in sorted(logs)) + " |")
        out.append("|---|---|" + "---|" * len(logs))
        rows = {}
        for f in sorted(logs):
            for s in logs[f].sites:
                if s["arith"] == ar and s["op"] not in CONST_OPS:
                    rows.setdefault((s["site"], s["op"]), {})[f] = s

        def site_key(k):
            fn, ln = k[0][0].rsplit(":", 1)
            return (fn, int(ln), k[0][1])

        for (site, op), per in sorted(rows.items(), key=site_key):
            cells = []
            for f in sorted(logs):
                s = per.get(f)
                if not s:
                    cells.append("")
                    continue
                L = logs[f].layout[ar]
                frac, what = load(s, L)
                cells.append("%s, %s -> %s; %s %.1f %%" % (fmt(s["a"] / L.p), fmt(s["b"] / L.p), fmt(s["out"] / L.p), what, 100 * frac))
            out.append("| `%s` | %s | %s |" % (site, op, " | ".join(cells)))
        out.append("")
    return "\n".join(out) + "\n"


def main():
    with open(os.path.join(ROOT, "tests", "golden", "params.json")) as f:
        params = json.load(f)
    logs = run(build())
    bad = judge(logs, params)
    steps = sum(len(l.steps) for l in logs.values())
    ops = sum(s["n"] for l in logs.values() for s in l.steps)
    print("%d operations walked, %d distinct steps judged, %d violations" % (ops, steps, len(bad)))
    for b in bad[:40]:
        print("  " + b)
    if "--table" in sys.argv:
        path = sys.argv[sys.argv.index("--table") + 1]
        with open(path, "w") as f:
            f.write(table(logs))
        print("wrote", path)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
