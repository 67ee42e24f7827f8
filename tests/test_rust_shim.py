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


# one table, three spellings of every parameter type that crosses the boundary: C header / ctypes binding / Rust -sys
def type_table():
    import ctypes as C
    from anemoi_amd import _lib
    P = C.POINTER
    return {
        "int": (C.c_int, "c_int"),
        "unsigned": (C.c_uint, "c_uint"),
        "unsigned int": (C.c_uint, "c_uint"),
        "size_t": (C.c_size_t, "usize"),
        "long long": (C.c_longlong, "c_longlong"),
        "long long *": (P(C.c_longlong), "*mut c_longlong"),
        "double *": (P(C.c_double), "*mut c_double"),
        "unsigned long long": (C.c_ulonglong, "c_ulonglong"),
        "int *": (P(C.c_int), "*mut c_int"),
        "uint64_t *": (P(C.c_uint64), "*mut u64"),
        "const uint64_t *": (P(C.c_uint64), "*const u64"),
        "uint8_t *": (P(C.c_uint8), "*mut u8"),
        "const uint8_t *": (P(C.c_uint8), "*const u8"),
        "const char *": (C.c_char_p, "*const c_char"),
        "void *": (C.c_void_p, "*mut c_void"),
        "const void *": (C.c_void_p, "*const c_void"),
        "const anemoi_generic_instance *": (P(_lib._GenericInstance), "*const AnemoiGenericInstance"),
        "anemoi_generic_handle *": (C.c_void_p, "*mut AnemoiGenericHandle"),
        "const anemoi_generic_handle *": (C.c_void_p, "*const AnemoiGenericHandle"),
        "anemoi_generic_handle * *": (P(C.c_void_p), "*mut *mut AnemoiGenericHandle"),
    }


def rust_extern_types(text):
    block = text[text.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (anemoi_[a-z0-9_]+)\((.*?)\)\s*->\s*([^;]+);", block, flags=re.S):
        args = [a.strip() for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = ([a.split(":", 1)[1].strip() for a in args], m.group(3).strip())
    return out


def test_parameter_types_agree_across_header_ctypes_and_rust():
    """Not only names and arities: every parameter's TYPE -- pointer constness (`const uint64_t*` <-> `*const u64`),
    `size_t` <-> `usize` <-> c_size_t, `unsigned` <-> `c_uint`, `uint8_t*` <-> `*mut u8`, the return types -- is the same
    in the C header, in the ctypes binding the GPU tests call through, and in the generated `extern "C"` block."""
    sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd"))
    from anemoi_amd import _lib
    funcs, _ = gen_rust_sys.parse_header()
    rust = rust_extern_types(open(LIB_RS).read())
    import ctypes
    TYPE_TABLE = type_table()
    ret_table = {"int": (ctypes.c_int, "c_int"), "const char *": (ctypes.c_char_p, "*const c_char"), "size_t": (ctypes.c_size_t, "usize")}
    for name, ret, params in funcs:
        cty_args, cty_ret = _lib._SIGS[name]
        r_args, r_ret = rust[name]
        assert (cty_ret, r_ret) == ret_table[ret], (name, ret)
        assert len(params) == len(cty_args) == len(r_args), name
        for (ctype, pname), ca, ra in zip(params, cty_args, r_args):
            want_ctypes, want_rust = TYPE_TABLE[ctype]
            assert ca is want_ctypes, (name, pname, ctype, ca.__name__)
            assert ra == want_rust, (name, pname, ctype, ra)
            if "const" in ctype and "*" in ctype:
                assert ra.startswith("*const"), (name, pname)     # inputs stay inputs on the Rust side
            if ctype.endswith("*") and "const" not in ctype and "char" not in ctype:
                assert ra.startswith("*mut"), (name, pname)


def test_patch_passes_felt_slices_the_way_the_header_reads_them():
    """The five symbols the reference patch calls take `&[Felt]` as `*const u64` / `*mut u64` (and bytes as `*const u8`),
    lengths as usize, field / width / k / device as c_int: the casts written in mi355x.rs must produce exactly those."""
    patch = open(PATCH).read()
    rust = rust_extern_types(open(LIB_RS).read())
    used = {"anemoi_jive_compress_k_batch": ["c_int", "c_int", "c_int", "*const u64", "*mut u64", "usize", "c_int"],
            "anemoi_hash_bytes_batch": ["c_int", "c_int", "*const u8", "usize", "usize", "*mut u64", "c_int"],
            "anemoi_hash_field_batch": ["c_int", "c_int", "*const u64", "usize", "usize", "*mut u64", "c_int"],
            "anemoi_merge_batch": ["c_int", "*const u64", "*mut u64", "usize", "c_int"],
            "anemoi_permutation_batch": ["c_int", "c_int", "*mut u64", "usize", "c_int"]}
    for name, want in used.items():
        assert rust[name][0] == want, (name, rust[name][0])
    assert "as_ptr() as *const u64" in patch and "as_mut_ptr() as *mut u64" in patch
    assert "as *const u8" in patch or "as_ptr()" in patch


def test_patch_uses_existing_symbols_and_covers_all_instances(params):
    patch = open(PATCH).read()
    ext = rust_externs(open(LIB_RS).read())
    used = re.findall(r"ffi::(anemoi_[a-z0-9_]+)\(([^;]*?)\)\s*\}\)", patch, flags=re.S)
    assert {u[0] for u in used} == {"anemoi_jive_compress_k_batch", "anemoi_hash_bytes_batch", "anemoi_hash_field_batch",
                                    "anemoi_merge_batch", "anemoi_permutation_batch", "anemoi_warmup"}
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
    for fn in ("compress_batch", "compress_k_batch", "hash_batch", "hash_field_batch", "merge_batch", "mi355x_warmup"):
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
