//! Raw bindings to the C-ABI declared in `include/anemoi_mi355x.h`.
//!
//! Element buffers are `*const u64` / `*mut u64`: `L` little-endian limbs per field element in
//! Montgomery form with R = 2^(64 L) -- the in-memory form of arkworks' `Fp<MontBackend<_, L>, L>`,
//! so a `&[Felt]` is passed as `slice.as_ptr() as *const u64` without conversion.
#![no_std]
#![allow(non_camel_case_types)]

use core::ffi::{c_char, c_int, c_uint, c_void};

pub const ANEMOI_BLS12_381: c_int = 0;
pub const ANEMOI_BLS12_377: c_int = 1;
pub const ANEMOI_BN_254: c_int = 2;
pub const ANEMOI_ED_ON_BLS12_377: c_int = 3;
pub const ANEMOI_JUBJUB: c_int = 4;
pub const ANEMOI_PALLAS: c_int = 5;
pub const ANEMOI_VESTA: c_int = 6;

pub const ANEMOI_ALL_DEVICES: c_int = -1;

pub const ANEMOI_OK: c_int = 0;
pub const ANEMOI_ERR_FIELD: c_int = -1;
pub const ANEMOI_ERR_WIDTH: c_int = -2;
pub const ANEMOI_ERR_ARG: c_int = -3;
pub const ANEMOI_ERR_DEVICE: c_int = -4;
pub const ANEMOI_ERR_ALLOC: c_int = -5;

/// `struct anemoi_generic_instance` of the header: NUM_COLUMNS, NUM_ROUNDS, ARK_C, ARK_D and an optional MDS
/// matrix (null = the hard-coded `mds_layer` arm, <= 6 columns) as Montgomery limbs, i.e. `&[Felt]` pointers.
#[repr(C)]
pub struct AnemoiGenericInstance {
    pub field: c_int,
    pub num_columns: c_int,
    pub num_rounds: c_int,
    pub ark_c: *const u64,
    pub ark_d: *const u64,
    pub mds: *const u64,
}

extern "C" {
    pub fn anemoi_abi_version() -> c_int;
    pub fn anemoi_device_count() -> c_int;
    pub fn anemoi_strerror(code: c_int) -> *const c_char;
    pub fn anemoi_last_error() -> *const c_char;
    pub fn anemoi_field_limbs(field: c_int) -> c_int;

    pub fn anemoi_permutation_batch(field: c_int, width: c_int, states: *mut u64, n: usize, device: c_int) -> c_int;
    pub fn anemoi_jive_compress_batch(field: c_int, width: c_int, input: *const u64, out: *mut u64, n: usize,
                                      device: c_int) -> c_int;
    pub fn anemoi_jive_compress_k_batch(field: c_int, width: c_int, k: c_int, input: *const u64, out: *mut u64,
                                        n: usize, device: c_int) -> c_int;
    pub fn anemoi_merge_batch(field: c_int, pairs: *const u64, out: *mut u64, n: usize, device: c_int) -> c_int;
    pub fn anemoi_hash_field_batch(field: c_int, width: c_int, elems: *const u64, elems_per_msg: usize, n: usize,
                                   out: *mut u64, device: c_int) -> c_int;
    pub fn anemoi_hash_bytes_batch(field: c_int, width: c_int, msgs: *const u8, msg_len: usize, n: usize,
                                   out: *mut u64, device: c_int) -> c_int;
    pub fn anemoi_merkle_root(field: c_int, leaves: *const u64, depth: c_uint, root: *mut u64, device: c_int) -> c_int;
    pub fn anemoi_merkle_tree(field: c_int, leaves: *const u64, depth: c_uint, tree: *mut u64, device: c_int) -> c_int;
    pub fn anemoi_merkle_path(field: c_int, tree: *const u64, depth: c_uint, index: usize, path: *mut u64) -> c_int;
    pub fn anemoi_merkle_verify_batch(field: c_int, leaves: *const u64, indices: *const u64, paths: *const u64,
                                      depth: c_uint, n: usize, root: *const u64, ok: *mut u8, device: c_int) -> c_int;
    pub fn anemoi_to_montgomery(field: c_int, input: *const u64, out: *mut u64, count: usize, device: c_int) -> c_int;
    pub fn anemoi_from_montgomery(field: c_int, input: *const u64, out: *mut u64, count: usize, device: c_int) -> c_int;

    // instances given by their `Anemoi` trait constants (src/traits.rs:36-76): what a new, wider instance binds
    pub fn anemoi_generic_mds_matrix(field: c_int, num_columns: c_int, mds: *mut u64, device: c_int) -> c_int;
    pub fn anemoi_generic_permutation_batch(inst: *const AnemoiGenericInstance, states: *mut u64, n: usize,
                                            device: c_int) -> c_int;
    pub fn anemoi_generic_jive_compress_k_batch(inst: *const AnemoiGenericInstance, k: c_int, input: *const u64,
                                                out: *mut u64, n: usize, device: c_int) -> c_int;
    pub fn anemoi_generic_hash_field_batch(inst: *const AnemoiGenericInstance, rate: c_int, elems: *const u64,
                                           elems_per_msg: usize, n: usize, out: *mut u64, device: c_int) -> c_int;
    pub fn anemoi_generic_hash_bytes_batch(inst: *const AnemoiGenericInstance, rate: c_int, msgs: *const u8,
                                           msg_len: usize, n: usize, out: *mut u64, device: c_int) -> c_int;
    pub fn anemoi_exp_alpha_batch(field: c_int, inverse: c_int, elems: *mut u64, n: usize, device: c_int) -> c_int;

    // buffers already in HBM; `stream` is a hipStream_t
    pub fn anemoi_jive_compress_k_dev(field: c_int, width: c_int, k: c_int, d_in: *const c_void, d_out: *mut c_void,
                                      n: usize, stream: *mut c_void) -> c_int;
    pub fn anemoi_hash_bytes_dev(field: c_int, width: c_int, d_msgs: *const c_void, msg_len: usize, n: usize,
                                 d_out: *mut c_void, stream: *mut c_void) -> c_int;
}
