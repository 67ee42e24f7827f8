#!/usr/bin/env python3
"""Per-kernel resources of the SHIPPED code objects: VGPRs, AGPRs, SGPRs, scratch, spills, LDS and the
waves per SIMD they allow -- read from the gfx950 code objects inside libanemoi_mi355x.so, not from the
source.  The register budget is a design point of this library (the 13-limb
kernels are built for 3 waves per SIMD, the 9-limb ones for 4), so it is gated: `check()` is what
tests/test_kernel_resources.py runs on the CPU, and the table is committed per round.

    python tools/kernel_resources.py                      # table on stdout
    python tools/kernel_resources.py --csv profiles/r03/kernel_resources.csv
    python tools/kernel_resources.py --lib path/to/other.so

How: the library's `.hip_fatbin` section is a sequence of clang offload bundles
(`__CLANG_OFFLOAD_BUNDLE__`, one per translation unit); each holds one `hipv4-amdgcn-amd-amdhsa--gfx950`
ELF whose NT_AMDGPU_METADATA note (msgpack, printed as YAML by `llvm-readelf --notes`) lists every kernel
with `.vgpr_count`, `.agpr_count`, `.sgpr_count`, `.private_segment_fixed_size` (scratch),
`.vgpr_spill_count`, `.sgpr_spill_count`, `.group_segment_fixed_size` (static LDS).  Dynamic LDS is not in the
code object; it is recomputed here from the same formula the launchers use (anemoi_kernels.h lds_bytes()).

Occupancy model (gfx950, /opt/skills/guides/MI355X_MICROARCH.md): 512 registers per lane per SIMD shared by
the architectural and accumulation files, allocated in blocks of 8 -> waves/SIMD = min(8, 512 // roundup8(vgpr +
agpr)); 160 KB of LDS per CU, workgroups of one wavefront -> waves/CU <= 163840 // lds_bytes.
"""
import argparse
import csv
import io
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "anemoi-rust_amd", "lib", "libanemoi_mi355x.so")
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = ["bls12_381", "bls12_377", "bn_254", "ed_on_bls12_377", "jubjub", "pallas", "vesta"]
LIMBS = {0: 13, 1: 13, 2: 9, 3: 9, 4: 9, 5: 9, 6: 9}          # lane-private limbs (30-bit / 29-bit)
ABI_WORDS = {0: 12, 1: 12, 2: 8, 3: 8, 4: 8, 5: 8, 6: 8}
LDS_PER_CU = 160 * 1024
WIN = 3


def fatbin_section(path):
    out = subprocess.check_output([os.path.join(LLVM_BIN, "llvm-readelf"), "-S", "-W", path], text=True)
    for line in out.splitlines():
        m = re.search(r"\.hip_fatbin\s+PROGBITS\s+[0-9a-f]+\s+([0-9a-f]+)\s+([0-9a-f]+)", line)
        if m:
            return int(m.group(1), 16), int(m.group(2), 16)
    raise RuntimeError("no .hip_fatbin section in %s" % path)


def code_objects(path):
    """the gfx950 ELF images of every offload bundle in the library"""
    off, size = fatbin_section(path)
    with open(path, "rb") as f:
        f.seek(off)
        blob = f.read(size)
    objs, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        (n,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            eoff, esize, tsize = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tsize].decode()
            q += 24 + tsize
            if "gfx950" in triple and esize:
                objs.append(blob[pos + eoff:pos + eoff + esize])
        pos += len(MAGIC)
    return objs


def demangle(names):
    """`k_jive<0, 2, 2>` from `_ZN6anemoi6k_jiveILi0ELi2ELi2EEEv...`: the kernels' template arguments are all
    integer / bool literals, so a ten-line Itanium decoder does (the image's LLVM ships no llvm-cxxfilt)"""
    out = []
    for n in names:
        m = re.match(r"_ZN6anemoi(\d+)", n) or re.match(r"_Z(\d+)", n)
        if not m:
            out.append(n)
            continue
        ln = int(m.group(1))
        base, rest = n[m.end():m.end() + ln], n[m.end() + ln:]
        args = []
        if rest.startswith("I"):
            rest = rest[1:]
            while True:
                a = re.match(r"L([ibjmlx])(n?\d+)E", rest)
                if not a:
                    break
                v = a.group(2).replace("n", "-")
                args.append({"0": "false", "1": "true"}[v] if a.group(1) == "b" else v)
                rest = rest[a.end():]
        out.append("anemoi::" + base + ("<" + ", ".join(args) + ">" if args else ""))
    return out


KEYS = {".vgpr_count": "vgpr", ".agpr_count": "agpr", ".sgpr_count": "sgpr",
        ".private_segment_fixed_size": "scratch", ".vgpr_spill_count": "vgpr_spills",
        ".sgpr_spill_count": "sgpr_spills", ".group_segment_fixed_size": "lds_static",
        ".max_flat_workgroup_size": "max_wg"}


def kernels_of(elf_bytes):
    import yaml
    with tempfile.NamedTemporaryFile(suffix=".co") as tf:
        tf.write(elf_bytes)
        tf.flush()
        notes = subprocess.check_output([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", tf.name], text=True)
    m = re.search(r"^\s*---\s*$(.*?)^\s*\.\.\.\s*$", notes, re.S | re.M)
    if not m:
        return []
    meta = yaml.safe_load(m.group(1))
    out = []
    for k in meta.get("amdhsa.kernels", []):
        rec = {"name": k[".name"], "symbol": k[".symbol"]}
        for key, short in KEYS.items():
            rec[short] = int(k.get(key, 0))
        out.append(rec)
    return out


def lds_dynamic(short, field):
    """the launchers' dynamic LDS request (anemoi_kernels.h lds_bytes<A, WIN, W>): max(window table, staging)"""
    nl, nabi = LIMBS[field], ABI_WORDS[field]
    nq = (nl + 3) // 4
    tab = ((1 << (WIN - 1)) - 1) * nq * 16 * 64
    if "_coop<" in short or short.startswith("k_assemble") or short.startswith("k_generic_prepare"):
        return 0
    if short.startswith("k_mont_convert"):
        return nabi * 4 * 64
    w = 1 if (short.startswith("k_sponge<") or short.startswith("k_sponge_ragged<") or short.startswith("k_merkle_climb")
              or short.startswith("k_exp_alpha") or "_cols" in short) else 2
    return max(tab, w * nabi * 4 * 64)


def short_name(dem):
    m = re.search(r"anemoi::(k_[a-z0-9_]+)(<[^>]*>)?", dem)
    if not m:
        m = re.search(r"(k_[a-z0-9_]+)", dem)
        return m.group(1) if m else dem
    return m.group(1) + (m.group(2) or "")


def field_of(short):
    m = re.search(r"<(\d+)", short)
    return int(m.group(1)) if m else None


def waves_per_simd(vgpr, agpr):
    regs = -(-(vgpr + agpr) // 8) * 8
    return min(8, 512 // max(regs, 8))


def collect(lib=DEFAULT_LIB):
    rows = []
    for obj in code_objects(lib):
        ks = kernels_of(obj)
        dem = demangle([k["name"] for k in ks])
        for k, d in zip(ks, dem):
            short = short_name(d)
            f = field_of(short)
            lds_dyn = lds_dynamic(short, f) if f is not None else 0
            lds = k.get("lds_static", 0) + lds_dyn
            wps = waves_per_simd(k.get("vgpr", 0), k.get("agpr", 0))
            wcu_lds = LDS_PER_CU // lds if lds else 32
            rows.append({
                "kernel": short, "field": FIELDS[f] if f is not None else "", "limbs": LIMBS.get(f, 0),
                "vgpr": k.get("vgpr", 0), "agpr": k.get("agpr", 0), "sgpr": k.get("sgpr", 0),
                "scratch": k.get("scratch", 0), "vgpr_spills": k.get("vgpr_spills", 0),
                "sgpr_spills": k.get("sgpr_spills", 0), "lds_static": k.get("lds_static", 0), "lds_dynamic": lds_dyn,
                "waves_per_simd_regs": wps, "waves_per_cu_lds": min(32, wcu_lds),
                "waves_per_cu": min(4 * wps, wcu_lds, 32),
            })
    rows.sort(key=lambda r: (r["field"], r["kernel"]))
    return rows


# ---- the gate ---------------------------------------------------------------------------------------
# Hot kernels = the fixed-instance throughput kernels.  Budgets (VGPR + AGPR): 13-limb fields 168 (3 waves per
# SIMD), 9-limb fields 128 (4 waves per SIMD).  EXCEPTIONS lists (kernel-prefix, field) pairs that are allowed a
# different budget, each with the measurement that justifies it.
HOT_PREFIXES = ("k_jive<", "k_jive_pair<", "k_sponge<", "k_sponge_pair<", "k_sponge_ragged<", "k_sponge_ragged_pair<",
                "k_permutation<", "k_permutation_pair<", "k_merkle_climb<", "k_exp_alpha<")
GENERIC_PREFIXES = ("k_permutation_cols<", "k_jive_cols<", "k_sponge_cols<")
EXCEPTIONS = {}   # (prefix, field name) -> (budget, "why")


def budget_for(row):
    k, fld = row["kernel"], row["field"]
    for (pre, f), (b, _) in EXCEPTIONS.items():
        if k.startswith(pre) and f == fld:
            return b
    if k.startswith(HOT_PREFIXES) or k.startswith(GENERIC_PREFIXES):
        return 168 if row["limbs"] >= 13 else 128
    return None


def check(rows):
    """-> list of violations (strings); empty = every kernel inside its budget, nothing spills to scratch"""
    bad = []
    for r in rows:
        # SGPR spills go to VGPR lanes (v_writelane), not to memory: reported in the table, not gated
        if r["scratch"] or r["vgpr_spills"]:
            bad.append("%s: scratch %d B, %d VGPR spills" % (r["kernel"], r["scratch"], r["vgpr_spills"]))
        b = budget_for(r)
        if b is not None and r["vgpr"] + r["agpr"] > b:
            bad.append("%s (%s): %d VGPRs + %d AGPRs > budget %d (%d waves/SIMD instead of %d)"
                       % (r["kernel"], r["field"], r["vgpr"], r["agpr"], b, r["waves_per_simd_regs"], 512 // b))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=DEFAULT_LIB)
    ap.add_argument("--csv")
    args = ap.parse_args()
    rows = collect(args.lib)
    cols = list(rows[0].keys()) + ["budget"]
    for r in rows:
        r["budget"] = budget_for(r) or ""
    buf = io.StringIO()
    w = csv.DictWriter(buf, fieldnames=cols, lineterminator="\n")
    w.writeheader()
    w.writerows(rows)
    if args.csv:
        os.makedirs(os.path.dirname(os.path.abspath(args.csv)), exist_ok=True)
        with open(args.csv, "w") as f:
            f.write(buf.getvalue())
        print("wrote %s (%d kernels)" % (args.csv, len(rows)))
    else:
        print("%-46s %-16s %5s %5s %5s %7s %6s %9s %6s" % ("kernel", "field", "vgpr", "agpr", "sgpr", "scratch",
                                                           "lds", "waves/SIMD", "w/CU"))
        for r in rows:
            print("%-46s %-16s %5d %5d %5d %7d %6d %9d %6d %s" % (
                r["kernel"][:46], r["field"], r["vgpr"], r["agpr"], r["sgpr"], r["scratch"],
                r["lds_static"] + r["lds_dynamic"], r["waves_per_simd_regs"], r["waves_per_cu"],
                ("<= %s" % r["budget"]) if r["budget"] else ""))
    bad = check(rows)
    for b in bad:
        print("OVER BUDGET:", b, file=sys.stderr)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
