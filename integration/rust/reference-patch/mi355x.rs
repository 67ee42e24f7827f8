// src/mi355x.rs -- NEW FILE in a fork of anemoi-hash/anemoi-rust, compiled only with `--features mi355x`
// (Cargo.toml: `mi355x = ["dep:anemoi-mi355x-sys", "std"]`; src/lib.rs: `#[cfg(feature = "mi355x")] mod mi355x;`).
//
// NOT BUILT IN THIS REPOSITORY'S IMAGE (no rustc).  tests/test_rust_shim.py checks that every FFI symbol
// used below exists in the generated -sys crate with the same arity, that all 14 instances are
// instantiated, and that the policy constant agrees with INTEGRATION.md.
//
// One macro adds BATCHED inherent functions to an instance's unit struct; the trait impls of
// src/<field>/anemoi_{2_1,4_3}/hasher.rs (`Sponge`, `Jive`, src/traits.rs:8-33) stay as they are.  The batch
// functions reduce, item by item, to the trait functions:
//     compress_batch(s)[i]        == <I as Jive>::compress(&s[W*i .. W*(i+1)])        (hasher.rs:96-103 / 4-3 :148-160)
//     compress_k_batch(s, k)[..]  == <I as Jive>::compress_k(.., k)                   (hasher.rs:105-110 / 4-3 :162-179)
//     hash_batch(msgs, len)[i]    == <I as Sponge>::hash(&msgs[len*i .. len*(i+1)])   (hasher.rs:18-66 / 4-3 :19-91)
//     hash_field_batch(e, m)[i]   == <I as Sponge>::hash_field(&e[m*i .. m*(i+1)])    (hasher.rs:68-85 / 4-3 :93-129)
//     merge_batch(pairs)[i]       == <I as Sponge>::merge(&pairs[i])                  (hasher.rs:86-92 / 4-3 :131-145)
//
// SMALL-BATCH POLICY (one rule for every function): fewer than MI355X_MIN_BATCH items stay on the CPU,
// item by item through the unchanged trait functions; MI355X_MIN_BATCH or more go to the GPU.  A single
// Jive::compress costs 1.71 ms (BLS12-381) / 0.96 ms (Jubjub) on the GPU's latency kernel against ~0.43 ms /
// ~0.13 ms on one CPU core (profiles/r04/reference_criterion_workloads.txt, reference README.md:77-78), so
// the GPU only pays from a few tens of items up; 32 is past break-even for all 14 instances.
// `Jive::compress` / `Sponge::hash` themselves (batches of one) are therefore NOT rerouted.
#![allow(unsafe_code)] // the crate is #![deny(unsafe_code)] (src/lib.rs:13): this is its one FFI module

#[cfg(not(feature = "std"))]
use alloc::vec::Vec;

use crate::{Jive, Sponge};
use anemoi_mi355x_sys as ffi;
use ark_ff::Zero;

/// Batches smaller than this run on the CPU through the reference's own functions.
pub const MI355X_MIN_BATCH: usize = 32;

/// The header's ABI-version rule (include/anemoi_mi355x.h): the library answers 100 x major + minor of the header IT was built
/// from; this crate, generated from a header of (ANEMOI_ABI_MAJOR, ANEMOI_ABI_MINOR), may use a library of the same major and
/// the same or a later minor, and nothing else.
pub fn mi355x_abi_compatible() -> bool {
    let v = unsafe { ffi::anemoi_abi_version() };
    v / 100 == ffi::ANEMOI_ABI_MAJOR && v % 100 >= ffi::ANEMOI_ABI_MINOR
}

/// The library never aborts: its error codes become the panics the reference raises itself.
#[inline]
fn check(rc: core::ffi::c_int) {
    assert!(rc == ffi::ANEMOI_OK, "anemoi_mi355x error {}", rc);
}

// $field / $shape are the reference's module names (src/lib.rs:27-64, src/<field>/mod.rs:8-12): `Felt` is the field
// module's public alias (src/<field>/mod.rs:1), the digest type and the sizes are the instance module's public items
// (src/<field>/anemoi_X_Y/mod.rs:13-31).
macro_rules! impl_mi355x {
    ($field:ident, $shape:ident, $inst:ident, $field_id:expr, $limbs:expr, $width:expr) => {
        const _: () = {
            use crate::$field::$shape::{AnemoiDigest, $inst, DIGEST_SIZE, STATE_WIDTH};
            use crate::$field::Felt;

            // `&[Felt]` is handed to C as `*const u64`: arkworks `Fp<MontBackend<_, L>, L>` is
            // `Fp(BigInt<L>([u64; L]), PhantomData)` -- L Montgomery limbs -- but neither type is
            // repr(C)/repr(transparent), so size and alignment are asserted at compile time.
            assert!(core::mem::size_of::<Felt>() == 8 * $limbs);
            assert!(core::mem::align_of::<Felt>() == 8);
            assert!(core::mem::size_of::<AnemoiDigest>() == 8 * $limbs * DIGEST_SIZE);
            assert!(STATE_WIDTH == $width && DIGEST_SIZE == 1);

            impl $inst {
                /// Once at start-up, for a service that cares about the duration of its FIRST large call: uploads this
                /// instance's constant tables to every GPU and runs each of its throughput kernels once on a small batch
                /// (~15 ms per GPU; include/anemoi_mi355x.h anemoi_warmup).  Changes no result.
                pub fn mi355x_warmup() {
                    assert!(mi355x_abi_compatible(), "libanemoi_mi355x.so has another ABI major than this crate was generated for");
                    check(unsafe { ffi::anemoi_warmup(ffi::ANEMOI_ALL_DEVICES, $field_id, $width) });
                }

                /// n states of STATE_WIDTH elements -> n x (STATE_WIDTH / 2) elements.
                pub fn compress_batch(states: &[Felt]) -> Vec<Felt> {
                    Self::compress_k_batch(states, 2)
                }

                /// n states -> n x (STATE_WIDTH / k) elements; k as `Jive::compress_k` accepts it.
                pub fn compress_k_batch(states: &[Felt], k: usize) -> Vec<Felt> {
                    assert!(states.len() % STATE_WIDTH == 0);
                    assert!(k != 0 && STATE_WIDTH % k == 0 && k % 2 == 0); // hasher.rs:107 / 4-3 :163-165
                    let n = states.len() / STATE_WIDTH;
                    if n < MI355X_MIN_BATCH {
                        return states
                            .chunks(STATE_WIDTH)
                            .flat_map(|s| <Self as Jive<Felt>>::compress_k(s, k))
                            .collect();
                    }
                    let mut out = vec![Felt::zero(); n * (STATE_WIDTH / k)];
                    check(unsafe {
                        ffi::anemoi_jive_compress_k_batch(
                            $field_id,
                            $width,
                            k as core::ffi::c_int,
                            states.as_ptr() as *const u64,
                            out.as_mut_ptr() as *mut u64,
                            n,
                            ffi::ANEMOI_ALL_DEVICES,
                        )
                    });
                    out
                }

                /// n messages of `msg_len` bytes each, contiguous -> n digests.
                pub fn hash_batch(msgs: &[u8], msg_len: usize) -> Vec<AnemoiDigest> {
                    let n = if msg_len == 0 { 0 } else { msgs.len() / msg_len };
                    assert!(msg_len == 0 || msgs.len() % msg_len == 0);
                    if n < MI355X_MIN_BATCH {
                        return msgs.chunks(msg_len.max(1)).map(<Self as Sponge<Felt>>::hash).collect();
                    }
                    let mut out = vec![AnemoiDigest::default(); n];
                    check(unsafe {
                        ffi::anemoi_hash_bytes_batch(
                            $field_id,
                            $width,
                            msgs.as_ptr(),
                            msg_len,
                            n,
                            out.as_mut_ptr() as *mut u64,
                            ffi::ANEMOI_ALL_DEVICES,
                        )
                    });
                    out
                }

                /// n messages of `elems_per_msg` field elements each, contiguous -> n digests.
                pub fn hash_field_batch(elems: &[Felt], elems_per_msg: usize) -> Vec<AnemoiDigest> {
                    let n = if elems_per_msg == 0 { 0 } else { elems.len() / elems_per_msg };
                    assert!(elems_per_msg == 0 || elems.len() % elems_per_msg == 0);
                    if n < MI355X_MIN_BATCH {
                        return elems
                            .chunks(elems_per_msg.max(1))
                            .map(<Self as Sponge<Felt>>::hash_field)
                            .collect();
                    }
                    let mut out = vec![AnemoiDigest::default(); n];
                    check(unsafe {
                        ffi::anemoi_hash_field_batch(
                            $field_id,
                            $width,
                            elems.as_ptr() as *const u64,
                            elems_per_msg,
                            n,
                            out.as_mut_ptr() as *mut u64,
                            ffi::ANEMOI_ALL_DEVICES,
                        )
                    });
                    out
                }

                /// n pairs of digests -> n digests, each == `Sponge::merge(&pairs[i])`.
                pub fn merge_batch(pairs: &[[AnemoiDigest; 2]]) -> Vec<AnemoiDigest> {
                    let n = pairs.len();
                    if n < MI355X_MIN_BATCH {
                        return pairs.iter().map(<Self as Sponge<Felt>>::merge).collect();
                    }
                    let mut out = vec![AnemoiDigest::default(); n];
                    if $width == 2 {
                        // 2-1: merge = Jive compress of [left, right] (anemoi_2_1/hasher.rs:86-92)
                        check(unsafe {
                            ffi::anemoi_merge_batch(
                                $field_id,
                                pairs.as_ptr() as *const u64,
                                out.as_mut_ptr() as *mut u64,
                                n,
                                ffi::ANEMOI_ALL_DEVICES,
                            )
                        });
                    } else {
                        // 4-3: the reference stores digests[0] in BOTH rate cells (anemoi_4_3/hasher.rs:136-138,
                        // an upstream defect, reported and deliberately preserved) and returns
                        // permutation(state)[0]: build exactly that state and permute on the GPU.
                        let mut st = vec![Felt::zero(); n * STATE_WIDTH];
                        for (s, p) in st.chunks_mut(STATE_WIDTH).zip(pairs) {
                            s[0] = p[0].as_elements()[0];
                            s[1] = p[0].as_elements()[0];
                        }
                        check(unsafe {
                            ffi::anemoi_permutation_batch(
                                $field_id,
                                $width,
                                st.as_mut_ptr() as *mut u64,
                                n,
                                ffi::ANEMOI_ALL_DEVICES,
                            )
                        });
                        for (o, s) in out.iter_mut().zip(st.chunks(STATE_WIDTH)) {
                            *o = AnemoiDigest::new([s[0]]);
                        }
                    }
                    out
                }
            }
        };
    };
}

// field ids = include/anemoi_mi355x.h; limbs = u64 limbs of `Felt`; one line per instance of src/lib.rs:27-64
#[cfg(feature = "bls12_381")]
impl_mi355x!(bls12_381, anemoi_2_1, AnemoiBls12_381_2_1, ffi::ANEMOI_BLS12_381, 6, 2);
#[cfg(feature = "bls12_381")]
impl_mi355x!(bls12_381, anemoi_4_3, AnemoiBls12_381_4_3, ffi::ANEMOI_BLS12_381, 6, 4);
#[cfg(feature = "bls12_377")]
impl_mi355x!(bls12_377, anemoi_2_1, AnemoiBls12_377_2_1, ffi::ANEMOI_BLS12_377, 6, 2);
#[cfg(feature = "bls12_377")]
impl_mi355x!(bls12_377, anemoi_4_3, AnemoiBls12_377_4_3, ffi::ANEMOI_BLS12_377, 6, 4);
#[cfg(feature = "bn_254")]
impl_mi355x!(bn_254, anemoi_2_1, AnemoiBn254_2_1, ffi::ANEMOI_BN_254, 4, 2);
#[cfg(feature = "bn_254")]
impl_mi355x!(bn_254, anemoi_4_3, AnemoiBn254_4_3, ffi::ANEMOI_BN_254, 4, 4);
#[cfg(feature = "ed_on_bls12_377")]
impl_mi355x!(ed_on_bls12_377, anemoi_2_1, AnemoiEdOnBls12_377_2_1, ffi::ANEMOI_ED_ON_BLS12_377, 4, 2);
#[cfg(feature = "ed_on_bls12_377")]
impl_mi355x!(ed_on_bls12_377, anemoi_4_3, AnemoiEdOnBls12_377_4_3, ffi::ANEMOI_ED_ON_BLS12_377, 4, 4);
#[cfg(feature = "jubjub")]
impl_mi355x!(jubjub, anemoi_2_1, AnemoiJubjub_2_1, ffi::ANEMOI_JUBJUB, 4, 2);
#[cfg(feature = "jubjub")]
impl_mi355x!(jubjub, anemoi_4_3, AnemoiJubjub_4_3, ffi::ANEMOI_JUBJUB, 4, 4);
#[cfg(feature = "pallas")]
impl_mi355x!(pallas, anemoi_2_1, AnemoiPallas_2_1, ffi::ANEMOI_PALLAS, 4, 2);
#[cfg(feature = "pallas")]
impl_mi355x!(pallas, anemoi_4_3, AnemoiPallas_4_3, ffi::ANEMOI_PALLAS, 4, 4);
#[cfg(feature = "vesta")]
impl_mi355x!(vesta, anemoi_2_1, AnemoiVesta_2_1, ffi::ANEMOI_VESTA, 4, 2);
#[cfg(feature = "vesta")]
impl_mi355x!(vesta, anemoi_4_3, AnemoiVesta_4_3, ffi::ANEMOI_VESTA, 4, 4);
