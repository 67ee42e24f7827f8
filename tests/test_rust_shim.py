"""The Rust side of the boundary is text (no rustc in the image), so what CAN be checked is checked:
  * anemoi-mi355x-sys/src/lib.rs is exactly what tools/gen_rust_sys.py generates from the header today,
    declares every header function with the same argument count, and every header constant;
  * reference-patch/mi355x.rs only calls FFI symbols that exist, with the right number of arguments,
    instantiates all 14 instances of the reference with the right field id / limb count / width, and states
    the same small-batch threshold as INTEGRATION.md."""
import os
import re
import sys

from conftest import FIELD_IDS, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_sys  # noqa: E402

LIB_RS = os.path.join(ROOT, "integration", "rust", "anemoi-mi355x-sys", "src", "lib.rs")
PATCH = os.path.join(ROOT, "integration", "rust", "reference-patch", "mi355x.rs")


def rust_externs(text):
    block = text[text.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (anemoi_[a-z0-9_]+)\((.*?)\)\s*->", block, flags=re.S):
        args = [a for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = len(args)
    return out


def test_sys_crate_is_the_generated_one_and_matches_the_header():
    funcs, consts = gen_rust_sys.parse_header()
    text = open(LIB_RS).read()
    assert text == gen_rust_sys.render(funcs, consts), "run tools/gen_rust_sys.py"
    ext = rust_externs(text)
    assert ext == {name: len(params) for name, _, params in funcs}
    assert len(ext) >= 44
    for name, val in consts:
        assert re.search(r"pub const %s: c_int = %d;" % (name, val), text), name


def test_header_parser_sees_every_symbol_the_ctypes_binding_has():
    sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
    from anemoi_amd import _lib
    funcs, _ = gen_rust_sys.parse_header()
    assert sorted(n for n, _, _ in funcs) == sorted(_lib._SIGS)
    for name, _, params in funcs:
        assert len(params) == len(_lib._SIGS[name][0]), name


def test_patch_uses_existing_symbols_and_covers_all_instances(params):
    patch = open(PATCH).read()
    ext = rust_externs(open(LIB_RS).read())
    used = re.findall(r"ffi::(anemoi_[a-z0-9_]+)\(([^;]*?)\)\s*\}\)", patch, flags=re.S)
    assert {u[0] for u in used} == {"anemoi_jive_compress_k_batch", "anemoi_hash_bytes_batch", "anemoi_hash_field_batch",
                                    "anemoi_merge_batch", "anemoi_permutation_batch"}
    for name, args in used:
        nargs = len([a for a in args.split(",") if a.strip()])
        assert ext[name] == nargs, (name, nargs, ext[name])
    inst = re.findall(r"impl_mi355x!\(([a-z0-9_]+), anemoi_(2_1|4_3), (\w+), ffi::(ANEMOI_[A-Z0-9_]+), (\d), (\d)\);", patch)
    assert len(inst) == 14
    lib = open(LIB_RS).read()
    seen = set()
    for field, shape, struct, const, limbs, width in inst:
        assert field in FIELD_IDS and (field, shape) not in seen
        seen.add((field, shape))
        assert int(re.search(r"pub const %s: c_int = (\d+);" % const, lib).group(1)) == FIELD_IDS.index(field)
        assert int(limbs) == params[field]["u64_limbs"] and int(width) == (2 if shape == "2_1" else 4)
        assert struct.endswith("_" + shape)
    for fn in ("compress_batch", "compress_k_batch", "hash_batch", "hash_field_batch", "merge_batch"):
        assert "pub fn %s(" % fn in patch
    assert "size_of::<Felt>() == 8 * $limbs" in patch and "align_of::<Felt>() == 8" in patch


def test_one_small_batch_policy_everywhere():
    patch = open(PATCH).read()
    thr = int(re.search(r"pub const MI355X_MIN_BATCH: usize = (\d+);", patch).group(1))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"MI355X_MIN_BATCH\s*=\s*(\d+)", doc)
    assert m and int(m.group(1)) == thr
    assert "batch of one" not in doc.lower() or "not rerouted" in doc.lower()


def test_rust_text_is_at_least_well_bracketed():
    """No rustc here: the cheapest syntax check there is -- (), [], {} balance outside strings, chars and comments."""
    for path in (LIB_RS, PATCH):
        text = open(path).read()
        text = re.sub(r"//[^\n]*", "", text)
        text = re.sub(r'"(?:\\.|[^"\\])*"', '""', text)
        stack, pairs = [], {")": "(", "]": "[", "}": "{"}
        for ch in text:
            if ch in "([{":
                stack.append(ch)
            elif ch in pairs:
                assert stack and stack.pop() == pairs[ch], path
        assert not stack, path
