// NOT BUILT HERE.  What changes in the reference's `src/bls12_381/anemoi_2_1/hasher.rs` (the other 13
// instances differ only by field id, limb count and width) when the cargo feature `mi355x` is on.
// The trait signatures of `src/traits.rs` are untouched; batched forms are new inherent functions.

#[cfg(feature = "mi355x")]
#[allow(unsafe_code)] // the crate is #![deny(unsafe_code)] (src/lib.rs:13): this is the one FFI module
mod mi355x {
    use super::{Felt, STATE_WIDTH};
    use anemoi_mi355x_sys as ffi;
    use ark_ff::Zero;

    /// n states of STATE_WIDTH elements -> n digests.  `Felt` = Fp<MontBackend<_, 6>, 6> is six
    /// Montgomery u64 limbs in memory, exactly the library's element encoding.
    pub(super) fn compress_batch(states: &[Felt]) -> Vec<Felt> {
        assert!(states.len() % STATE_WIDTH == 0);
        let n = states.len() / STATE_WIDTH;
        let mut out = vec![Felt::zero(); n];
        let rc = unsafe {
            ffi::anemoi_jive_compress_batch(
                ffi::ANEMOI_BLS12_381,
                STATE_WIDTH as i32,
                states.as_ptr() as *const u64,
                out.as_mut_ptr() as *mut u64,
                n,
                ffi::ANEMOI_ALL_DEVICES,
            )
        };
        // the library never aborts; its error codes become the panics the reference raises itself
        assert!(rc == ffi::ANEMOI_OK, "anemoi_mi355x error {}", rc);
        out
    }
}

#[cfg(feature = "mi355x")]
impl AnemoiBls12_381_2_1 {
    /// Batched Jive compression: item i equals `Self::compress(&states[2 * i..2 * i + 2])[0]`.
    pub fn compress_batch(states: &[Felt]) -> Vec<Felt> {
        mi355x::compress_batch(states)
    }
}

impl Jive<Felt> for AnemoiBls12_381_2_1 {
    fn compress(elems: &[Felt]) -> Vec<Felt> {
        assert!(elems.len() == STATE_WIDTH); // hasher.rs:97, unchanged

        #[cfg(feature = "mi355x")]
        {
            // a batch of one costs a few milliseconds on the GPU against ~0.4 ms on a CPU core
            // (profiles/r01/reference_criterion_workloads.txt): single calls stay on the CPU,
            // callers with many states use compress_batch.
        }

        reference_cpu_compress(elems) // the unchanged body of hasher.rs:99-102, moved into a helper
    }

    fn compress_k(elems: &[Felt], k: usize) -> Vec<Felt> {
        assert!(k == 2); // hasher.rs:107, unchanged
        Self::compress(elems)
    }
}
