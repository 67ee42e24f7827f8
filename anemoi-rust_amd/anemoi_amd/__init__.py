"""anemoi_amd -- Python face of libanemoi_mi355x.so (ctypes over the C-ABI in include/anemoi_mi355x.h).

Mirrors the reference crate's operator surface for this path (src/traits.rs:8-33):
`Sponge::{hash, hash_field, merge}` and `Jive::{compress, compress_k}` as static-style methods of an
`Anemoi(field, width)` instance object (the reference's unit structs, e.g. `AnemoiBls12_381_2_1`),
plus the batched forms the GPU needs.  Where the reference panics (`assert!`), these raise
`AnemoiError`.  There is NO CPU fallback: if the HIP library cannot be loaded, importing fails.

Elements cross this boundary as numpy uint64 rows of `limbs` little-endian words in Montgomery
form (the arkworks in-memory form, see the header).  `to_montgomery` / `from_montgomery` convert
canonical integers on the GPU.
"""
from . import synth  # noqa: F401  (seeded workloads of the BASELINE configs)
from ._lib import (ALL_DEVICES, FIELD_IDS, Anemoi, AnemoiError, GenericAnemoi, builtin_mds_matrix, exp_alpha_batch, device_count, field_id, init, lib, lib_path, release, warmup, probe_issue_rate, ClockSampler, kernel_Mcycles,
                   AUTO, OPTIONS, options, set_option, get_option, is_ab_build,
                   from_montgomery, to_montgomery, ints_to_limbs, limbs_to_ints)

__all__ = ["ALL_DEVICES", "FIELD_IDS", "Anemoi", "AnemoiError", "GenericAnemoi", "builtin_mds_matrix",
           "exp_alpha_batch", "device_count", "field_id", "init", "lib", "lib_path", "release", "warmup", "probe_issue_rate", "ClockSampler", "kernel_Mcycles", "synth",
           "from_montgomery", "to_montgomery", "ints_to_limbs", "limbs_to_ints", "AUTO", "OPTIONS", "options", "is_ab_build",
           "set_option", "get_option"]
