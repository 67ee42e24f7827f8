#!/usr/bin/env python3
"""VGPR liveness of one kernel of the shipped library, from its disassembly: where is the register peak and
what is live there?  Companion of tools/kernel_resources.py (which says HOW MANY registers a kernel takes);
this says WHY, which is what one needs to bring a kernel back inside its budget.

    python tools/vgpr_liveness.py 'k_jive<1, 2, 2>'            # peak, and the live ranges that cross it
    python tools/vgpr_liveness.py 'k_jive<1, 2, 2>' --profile   # live-register count along the kernel

Method: llvm-objdump of the gfx950 code object, one basic-block graph from the s_branch / s_cbranch targets,
classic backward data-flow over VGPR numbers.  Operand roles are approximated (first vector operand of a
non-store instruction is its definition; DPP / partial writes are treated as full definitions), which is exact
enough for a peak: the kernels are straight-line big-integer code.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr  # noqa: E402

NO_DEF = ("ds_write", "global_store", "buffer_store", "flat_store", "scratch_store", "s_", "v_cmp", "v_cmpx",
          "v_readlane", "v_readfirstlane", "ds_bpermute_b32_nodef", "global_atomic", "v_nop", "ds_nop")


def vregs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return [int(m.group(1))] if m else []


def parse(asm_lines):
    ins = []
    for l in asm_lines:
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if not m:
            continue
        op, args, addr = m.group(1), m.group(2), int(m.group(3), 16)
        toks = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", args)] if args else []
        toks = [t.split(" ")[0] for t in toks]  # drop modifiers glued to the last operand
        extra = re.findall(r"\bv\[\d+:\d+\]|\bv\d+\b", args)
        defs, uses = [], []
        has_def = not op.startswith(NO_DEF)
        first = True
        for t in toks:
            r = vregs(t)
            if not r:
                first = False if t and not t.startswith(("vcc", "s[", "s", "exec")) else first
                continue
            if first and has_def:
                defs += r
                first = False
            else:
                uses += r
                first = False
        # registers mentioned inside modifiers (e.g. offen addressing) count as uses
        for t in extra:
            for r in vregs(t):
                if r not in defs and r not in uses:
                    uses.append(r)
        if op.startswith("v_swap"):
            uses += defs
        tgt = None
        mb = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", l)
        ins.append({"op": op, "args": args, "addr": addr, "defs": set(defs), "uses": set(uses), "text": l.strip(),
                    "branch": op.startswith(("s_branch", "s_cbranch")), "uncond": op == "s_branch",
                    "end": op == "s_endpgm", "tgt_off": int(mb.group(1), 16) if mb else tgt})
    return ins


def liveness(ins, base):
    addr_to_idx = {i["addr"]: k for k, i in enumerate(ins)}
    succ = []
    for k, i in enumerate(ins):
        s = []
        if i["end"]:
            succ.append(s)
            continue
        if i["branch"] and i["tgt_off"] is not None:
            t = addr_to_idx.get(base + i["tgt_off"])
            if t is not None:
                s.append(t)
        if not i["uncond"] and k + 1 < len(ins):
            s.append(k + 1)
        succ.append(s)
    live_in = [set() for _ in ins]
    changed = True
    while changed:
        changed = False
        for k in range(len(ins) - 1, -1, -1):
            out = set()
            for s in succ[k]:
                out |= live_in[s]
            new = (out - ins[k]["defs"]) | ins[k]["uses"]
            if new != live_in[k]:
                live_in[k] = new
                changed = True
    return live_in


def disassemble(kernel, lib):
    for obj in kr.code_objects(lib):
        ks = kr.kernels_of(obj)
        names = kr.demangle([k["name"] for k in ks])
        for k, d in zip(ks, names):
            if kr.short_name(d) == kernel:
                with tempfile.NamedTemporaryFile(suffix=".co") as tf:
                    tf.write(obj)
                    tf.flush()
                    out = subprocess.check_output([os.path.join(kr.LLVM_BIN, "llvm-objdump"), "-d", "--mcpu=gfx950",
                                                   "--disassemble-symbols=" + k["name"], tf.name], text=True)
                return out.splitlines(), k
    raise SystemExit("kernel %r not found" % kernel)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel")
    ap.add_argument("--lib", default=kr.DEFAULT_LIB)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--context", type=int, default=6)
    a = ap.parse_args()
    lines, meta = disassemble(a.kernel, a.lib)
    hdr = next(l for l in lines if re.match(r"^[0-9a-f]+ <", l))
    base = int(hdr.split()[0], 16)
    ins = parse(lines)
    live = liveness(ins, base)
    counts = [len(s) for s in live]
    peak = max(range(len(ins)), key=lambda k: counts[k])
    print("%s: %d instructions, .vgpr_count %d, max simultaneously live %d at instruction %d (%s)"
          % (a.kernel, len(ins), meta["vgpr"], counts[peak], peak, ins[peak]["text"][:60]))
    if a.profile:
        step = max(1, len(ins) // 120)
        for k in range(0, len(ins), step):
            print("%6d %4d %s" % (k, max(counts[k:k + step]), "#" * (max(counts[k:k + step]) // 3)))
        return
    # live ranges crossing the peak: for each register, where was it defined last before / used next after
    regs = sorted(live[peak])
    print("live at the peak: %d registers" % len(regs))
    groups, cur = [], []
    info = {}
    for r in regs:
        d = next((k for k in range(peak - 1, -1, -1) if r in ins[k]["defs"]), None)
        u = next((k for k in range(peak, len(ins)) if r in ins[k]["uses"]), None)
        info[r] = (d, u)
    for r in regs:
        d, u = info[r]
        print("  v%-3d def@%-6s %-44s next use@%-6s %s" % (r, d, ins[d]["text"][:44] if d is not None else "-",
                                                         u, ins[u]["text"][:50] if u is not None else "-"))


if __name__ == "__main__":
    main()
