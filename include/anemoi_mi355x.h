/*
 * anemoi_mi355x.h -- C-ABI of libanemoi_mi355x.so: batched Anemoi permutation, Jive compression and
 * sponge hashing on AMD Instinct MI355X (gfx950), hand-written HIP kernels.
 *
 * This is the drop-in boundary for the reference crate anemoi-hash/anemoi-rust: the reference has
 * no FFI and no batch entry point (every function processes ONE state, `#![deny(unsafe_code)]`,
 * src/lib.rs:13), so each entry point below is the batched form of one reference function and is
 * defined to equal, element by element, that function applied to each item.  A Rust `-sys` shim
 * binds these symbols (INTEGRATION.md shows it) underneath the unchanged `Sponge` / `Jive` traits
 * (src/traits.rs:8-33).
 *
 * ELEMENT ENCODING (all `uint64_t*` element buffers): one field element = L little-endian u64
 * limbs (L = anemoi_field_limbs(field): 6 for bls12_381 / bls12_377, 4 otherwise), in MONTGOMERY
 * form with R = 2^(64 L), fully reduced (< p).  That is byte-for-byte the in-memory form of the
 * reference's `Felt` = arkworks `Fp<MontBackend<_, L>, L>` (src/<field>/mod.rs:1-3), so a Rust
 * caller passes `slice.as_ptr()` unchanged.  Non-Rust callers convert canonical integers with
 * anemoi_to_montgomery / anemoi_from_montgomery.
 * UNREDUCED INPUTS.  arkworks keeps a `Felt` below p, so a Rust caller never sends anything else; a C caller's buffer
 * holds whatever words it holds.  The contract: ANY 64 L-bit pattern X in an element buffer is taken as X mod p (up to
 * 13.7 p on ed_on_bls12_377, 152 p on bls12_377) -- every entry point returns exactly what it returns for the reduced
 * value, and every OUTPUT element is fully reduced.  This is proven, not assumed: the walk of the kernels' lazy-reduction
 * bounds starts every ABI input at 2^(64 L) - 1 (csrc/BOUNDS.md, tests/test_bounds_walk.py), and
 * tests/test_gpu_parity.py::test_unreduced_abi_inputs_are_taken_mod_p feeds p, p + 1, 2^(64 L) - 1 ... to every entry
 * point under every kernel routing.  Two places hand the caller's BYTES through instead of a value, and say so below:
 * level 0 of a retained Merkle tree (and a depth-0 root) is a copy of the leaves as given; the `root` argument of the
 * verify functions is compared bytewise with a canonical value, so it must be reduced (every root the library returns is).
 *
 * OWNERSHIP / THREADING: the caller allocates and owns every buffer.  The library owns only its
 * per-device constant tables and a pool of "lanes" (three non-blocking HIP streams, reusable device
 * buffers and pinned staging each), created lazily under a mutex and kept until anemoi_release().
 * Every function is re-entrant and may be called from any thread: each host-pointer call borrows its
 * own lane, so concurrent callers run on their own streams and never on the NULL stream.  No function
 * aborts or throws: errors are negative return codes (the reference's `assert!` panics, e.g.
 * hasher.rs:97,107, map to ANEMOI_ERR_ARG and are re-raised as panics by the Rust shim).
 *
 * DEVICES: host-pointer functions take `device` = a HIP ordinal, or ANEMOI_ALL_DEVICES to split
 * the batch into contiguous ranges over every visible GPU, one host thread per GPU (no collective:
 * items are independent).  Large batches are cut into chunks of whole "waves of workgroups" and the
 * copy of chunk i + 1 runs under the kernel of chunk i, so the device footprint of a host-pointer call is
 * three chunks (~100 MB), not the batch.  The Merkle functions shard into one subtree per GPU (binary:
 * the largest power of two <= #GPUs; arity 4: the largest power of four) and finish the top levels on
 * the first device -- the subtree roots are the only cross-GPU data.
 * `_dev` functions take pointers already resident in the current device's HBM plus a hipStream_t
 * (passed as void*, NULL = default stream), enqueue asynchronously and do not synchronise -- except
 * that the FIRST use of a (device, field, width) uploads that instance's constant tables with a blocking
 * copy (not legal inside a stream capture): call anemoi_init() beforehand to get that out of the way.
 * Input and output ranges of a `_dev` Jive call must not overlap (ANEMOI_ERR_ARG).
 * STREAM CAPTURE: once its instance is initialised, every `_dev` function only launches kernels (a depth-0 tree: one
 * asynchronous device-to-device copy) on the stream it is handed, so it may be captured into a hipGraph; the options below
 * are read on the HOST when the call is made, so a graph keeps the kernels chosen at capture time whatever the options
 * become later (tests/test_gpu_capture.py).  No `_dev` function issues a hipMemsetAsync: on ROCm 7.2 a captured memset
 * node takes effect on the first replay only (profiles/r05/captured_memset_node_replays.txt).  Not capturable:
 * anemoi_init / anemoi_warmup / anemoi_release / anemoi_probe_issue_rate, anemoi_clock_sampler_*, anemoi_generic_prepare /
 * _destroy, and every host-pointer function (they copy and wait).
 *
 * OPTIONS: the kernel-selection cut-offs and the test / diagnostic knobs are listed with anemoi_set_option below.
 * Each is read from its environment variable ONCE (first use) and changed afterwards only through the API; no entry
 * point calls getenv() when it launches.
 */
#ifndef ANEMOI_MI355X_H
#define ANEMOI_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* field ids = the reference's module names (src/lib.rs:27-64) */
enum {
  ANEMOI_BLS12_381 = 0,       /* ark_bls12_381::Fq, 6 limbs, 47-byte chunks */
  ANEMOI_BLS12_377 = 1,       /* ark_bls12_377::Fq, 6 limbs, 47-byte chunks */
  ANEMOI_BN_254 = 2,          /* ark_bn254::Fq,     4 limbs, 31-byte chunks */
  ANEMOI_ED_ON_BLS12_377 = 3, /* ark_bls12_377::Fr, 4 limbs, alpha = 11, g = 22 */
  ANEMOI_JUBJUB = 4,          /* ark_bls12_381::Fr, 4 limbs */
  ANEMOI_PALLAS = 5,          /* ark_pallas::Fq,    4 limbs */
  ANEMOI_VESTA = 6,           /* ark_pallas::Fr,    4 limbs */
  ANEMOI_NUM_FIELDS = 7
};

/* width = STATE_WIDTH of the instance: 2 (Anemoi-2-1, rate 1) or 4 (Anemoi-4-3, rate 3) */

#define ANEMOI_ALL_DEVICES (-1)

#define ANEMOI_OK 0
#define ANEMOI_ERR_FIELD (-1)  /* unknown field id */
#define ANEMOI_ERR_WIDTH (-2)  /* width is not 2 or 4 */
#define ANEMOI_ERR_ARG (-3)    /* null pointer, unsupported k, depth out of range (reference: assert!) */
#define ANEMOI_ERR_DEVICE (-4) /* no such device, or a HIP call failed: see anemoi_last_error() */
#define ANEMOI_ERR_ALLOC (-5)  /* device or host allocation failed */

/* ---- introspection ----------------------------------------------------------------------
 * ABI VERSION RULE.  anemoi_abi_version() = 100 x ANEMOI_ABI_MAJOR + ANEMOI_ABI_MINOR of the header the LIBRARY was built
 * from.  MAJOR moves when an existing function changes its signature, its buffer layout or the meaning of an argument or of
 * a result (a caller must be re-read and re-compiled); MINOR moves, and MAJOR stays, when functions, options or error codes
 * are only ADDED.  A caller compiled against (M, m) may use a library that answers 100 x M + m' with m' >= m, and nothing
 * else.  History: 4 (56 functions); 11 functions were added in round 5 without moving it (by this rule: 4.1); 5.0 = the two
 * `_ragged_bucketed_dev` forms take the extent of the message blob and leave a status word in the scratch, whose size grew
 * by 16 bytes; first version with the options balance_underfilled and lane_priorities (67 functions). */
#define ANEMOI_ABI_MAJOR 5
#define ANEMOI_ABI_MINOR 0
int anemoi_abi_version(void);
int anemoi_device_count(void);               /* visible HIP devices, or a negative error */
const char *anemoi_strerror(int code);
const char *anemoi_last_error(void);         /* thread-local detail of the last ANEMOI_ERR_DEVICE */
int anemoi_field_id(const char *name);       /* "bls12_381" -> 0 ...; ANEMOI_ERR_FIELD if unknown */
const char *anemoi_field_name(int field);
int anemoi_field_limbs(int field);           /* u64 limbs per element */
int anemoi_field_chunk_bytes(int field);     /* 31 or 47: bytes absorbed per element by hash() */
int anemoi_num_rounds(int field, int width); /* NUM_HASH_ROUNDS, src/<f>/anemoi_x/mod.rs:31 */

/* ---- lifecycle ------------------------------------------------------------------------------
 * Optional.  anemoi_init uploads the constant tables of (field, width) to `device` (or to every device
 * with ANEMOI_ALL_DEVICES) and creates one lane, so that later calls -- `_dev` calls in particular --
 * neither allocate nor synchronise.  anemoi_release frees everything the library holds on `device`
 * (constant tables, idle lanes with their streams and buffers); ANEMOI_ERR_ARG while host-pointer calls are in
 * flight there.  `_dev` work the caller has enqueued on its own streams is the caller's to wait for first: its
 * kernels read the constant tables -- and a hipGraph captured from `_dev` calls holds the tables' addresses: capture again
 * after a release, do not replay.  The library can be used again afterwards (it re-initialises lazily). */
int anemoi_init(int device, int field, int width);
int anemoi_release(int device);
/* anemoi_init, then ONE small launch (16 x SIMDs items of zeros, a few ms) of every throughput kernel of (field, width):
 * Jive, permutation, sponge over bytes and over elements.  Why: the first big dispatch of a process can place its
 * wavefronts unevenly over the SIMDs of a CU (three on one, one on another); a launch whose workgroups all start at
 * once and stay for its whole duration -- 2^16 long messages: 2 048 wavefronts for a third of a second -- then takes
 * x 1.5 (profiles/r05/).  A service that cares about its FIRST large call calls this at start-up;
 * it waits for its launches, so it is not for capture either.  (Since round 6 the library itself precedes every launch that
 * does not fill the chip by a do-nothing launch that restores an even placement -- option balance_underfilled -- which covers
 * the first call as well; anemoi_warmup still takes the constant upload and the code-object load out of that call.) */
int anemoi_warmup(int device, int field, int width);

/* ---- diagnostics ----------------------------------------------------------------------------
 * What this GPU delivers right now of the instruction that carries the throughput kernels, two fixed kernels (~20 ms and
 * ~200 ms) on a full grid (three wavefronts per SIMD) of `device`:
 *   (1) bare dependent v_mad_u64_u32 chains      -> *lane_mad_per_s (lane multiply-adds per second; the ceiling at 16
 *       lanes per clock is SIMDs x 16 x clock) and *shader_clock_ghz, the clock the chip held meanwhile (s_memtime /
 *       s_memrealtime, median over the workgroups);
 *   (2) a chain of the generated BLS12-381 squaring (260 multiply-adds among 345 instructions: the instruction mix and
 *       power draw of the headline kernel)       -> *sqr_lane_mad_per_s, *sqr_shader_clock_ghz.
 * bench.py reports them beside its line, so that a slow box and a slow build can be told apart: boxes of one pool differ
 * by several per cent under this load, a kernel's rate as a fraction of probe (2) does not. */
int anemoi_probe_issue_rate(int device, double *lane_mad_per_s, double *shader_clock_ghz, double *sqr_lane_mad_per_s,
                            double *sqr_shader_clock_ghz);

/* The shader clock the chip HOLDS WHILE the caller's own kernels run -- what differs from box to box under this load, and
 * what a process cannot otherwise read.  A sampler of 16 single-wavefront workgroups (one wave slot and a handful of
 * registers each, asleep between samples, spread over the XCDs) is started on a stream of its own and logs (wall clock,
 * shader-cycle counter) every period_us; the caller brackets its work with two wall-clock stamps on ITS stream, stops the
 * sampler and reads the mean / min / max clock (GHz, over the sampler's workgroups) between the stamps:
 *     buf = device memory of anemoi_clock_sampler_bytes(), 8-byte aligned; stamps = two device uint64
 *     anemoi_clock_sampler_start_dev(buf, bytes, 2000, 60000, side_stream);   anemoi_clock_sampler_wait_dev(buf, 50, work_stream);
 *                                                                              anemoi_clock_stamp_dev(&stamps[0], work_stream);
 *     ... the work, on work_stream ...                                         anemoi_clock_stamp_dev(&stamps[1], work_stream);
 *     anemoi_clock_sampler_stop_dev(buf, third_stream);   synchronise;   copy buf and stamps to the host;
 *     anemoi_clock_sampler_read(host_buf, bytes, stamps[0], stamps[1], &mean, &lo, &hi, &groups);
 * (period_us 10 ... 1 000 000, max_ms 1 ... 600 000, else ANEMOI_ERR_ARG.)
 * The sampler ends when stopped, when its log (4 096 samples) is full, or after max_ms, whichever comes first.  The stop
 * must be issued on a stream that does not wait for the sampler.  NEITHER may the work's or the stop's stream share a HARDWARE
 * QUEUE with the sampler's: HIP multiplexes the streams of one priority onto GPU_MAX_HW_QUEUES (4) queues, round-robin in
 * the order they are created, and runs the kernels of streams that share a queue one after the other -- the work would wait
 * until the sampler's log is full (4 096 periods) and then run without a sampler.  Streams of different priorities never
 * share a queue: create side_stream with hipStreamCreateWithPriority(.., hipStreamNonBlocking, <greatest priority>) -- the
 * sampler sleeps, its priority costs the work nothing -- and keep the work and the stop on default-priority streams
 * (profiles/r06/sampler_queue_collision.txt; anemoi_amd.ClockSampler does this).  anemoi_clock_sampler_wait_dev holds `stream` (a one-lane
 * kernel, at most timeout_ms = 1 ... 10 000) until the sampler has taken its first sample: a sampler whose stream is slow
 * to start would otherwise miss a short piece of work.  bench.py reports the clock of its timed steps this way. */
size_t anemoi_clock_sampler_bytes(void);
int anemoi_clock_sampler_start_dev(void *d_buf, size_t bytes, unsigned period_us, unsigned max_ms, void *stream);
int anemoi_clock_sampler_wait_dev(void *d_buf, unsigned timeout_ms, void *stream);
int anemoi_clock_sampler_stop_dev(void *d_buf, void *stream);
int anemoi_clock_stamp_dev(void *d_u64, void *stream);
int anemoi_clock_sampler_read(const void *h_buf, size_t bytes, unsigned long long t0, unsigned long long t1, double *ghz_mean,
                              double *ghz_min, double *ghz_max, int *groups_used);

/* ---- options --------------------------------------------------------------------------------
 * anemoi_set_option(name, value): value -1 = automatic (the default).  `name` is the option name or its environment
 * variable.  Each option starts from its environment variable, read and validated ONCE at first use (a value that is
 * not a whole number in range is ignored; anemoi_last_error() after anemoi_get_option says so).  Unknown name or
 * value out of range: ANEMOI_ERR_ARG.  Changing an option affects calls that START afterwards.
 *
 *   name                   environment                  automatic value                meaning
 *   coop2d_max             ANEMOI_COOP2D_MAX            2 x SIMDs of the device        largest 2-1 batch (Jive / merge, permutation, sponge, path
 *                                                                                      climb) on the two-row 2-D latency kernels (two items per
 *                                                                                      wavefront, lowest latency); 0 = never
 *   coop4_max              ANEMOI_COOP4_MAX             8 x SIMDs                      largest Jive 2-1 / permutation batch on the row-cooperative
 *                                                                                      kernel (four items per wavefront); above: lane-private
 *   coop43_max             ANEMOI_COOP43_MAX            4 x SIMDs                      the same for Anemoi-4-3 (two states per wavefront)
 *   coop2d43_max           ANEMOI_COOP2D43_MAX          1 x SIMDs                      largest Anemoi-4-3 batch (Jive, permutation, sponge) on the
 *                                                                                      two-row 2-D kernels (ONE state per wavefront, a column per
 *                                                                                      row pair: lowest latency); above: coop43_max's kernel
 *   coop_sponge_max        ANEMOI_COOP_SPONGE_MAX       4 x SIMDs                      largest equal-length sponge batch on the cooperative kernel
 *   coop_climb_max         ANEMOI_COOP_CLIMB_MAX        4 x SIMDs                      largest batch of authentication paths on the cooperative kernel
 *   (Laboratory libraries built with `make AB=1` also know coop_max / ANEMOI_COOP_MAX: the one-item-per-wavefront kernels that
 *   measured slower and are not compiled into the product.  The product answers ANEMOI_ERR_ARG for that name.)
 *   virtual_devices        ANEMOI_VIRTUAL_DEVICES       the GPU count                  ANEMOI_ALL_DEVICES shards into this many ranges / subtrees,
 *                                                                                      mapped round-robin onto the GPUs (multi-GPU code on one GPU)
 *   host_staging           ANEMOI_HOST_STAGING          1 ("pinned")                   1: host buffers go through the lane's pinned staging;
 *                                                                                      0 ("direct"): copied straight from / to the caller's memory
 *   chunk_target_bytes     ANEMOI_CHUNK_TARGET_BYTES    24 MiB                         input bytes per chunk of the host pipelines (test knob)
 *   test_quantum           ANEMOI_TEST_QUANTUM          occupancy API                  items per full wave of workgroups (test knob)
 *   sponge_segment_bytes   ANEMOI_SPONGE_SEGMENT_BYTES  by batch shape                 forces the segment-fed sponge, this many bytes per segment
 *   balance_underfilled    ANEMOI_BALANCE_UNDERFILLED   1                              1: a launch of 2 ... 16 single-wavefront workgroups per CU (one
 *                                                                                      that does not fill the chip) is preceded by a do-nothing launch
 *                                                                                      of 4 workgroups per CU (~25 us): without it such a launch takes
 *                                                                                      x 1.3 ... 1.9 whenever it follows a launch that over-filled the
 *                                                                                      chip -- the dispatcher stacks its wavefronts on some SIMDs
 *                                                                                      (profiles/r06/underfilled_launch_placement.txt); 0: off (A/B)
 *   lane_priorities        ANEMOI_LANE_PRIORITIES       1                              1: every second lane's kernel stream (and the second kernel stream
 *                                                                                      of every lane's chunked pipeline) is a HIGH-priority stream: HIP keeps
 *                                                                                      a set of hardware queues per priority, so concurrent callers spread
 *                                                                                      over twice the queues (read when a lane is created); 0: round 5
 */
int anemoi_set_option(const char *name, long long value);
int anemoi_get_option(const char *name, long long *value); /* the value in force; -1 = automatic */

/* ---- host-pointer batch API -------------------------------------------------------------- */

/* Anemoi::permutation (src/traits.rs:370-378) on n states of `width` elements, in place. */
int anemoi_permutation_batch(int field, int width, uint64_t *states, size_t n, int device);

/* Anemoi::sbox_layer (src/traits.rs:326-358) on n states, in place: the unit pinned by the
 * reference's test_sbox vectors (src/<f>/anemoi_x/mod.rs:68). */
int anemoi_sbox_layer_batch(int field, int width, uint64_t *states, size_t n, int device);

/* Jive::compress (src/<f>/anemoi_2_1/hasher.rs:96-103, anemoi_4_3/hasher.rs:148-160):
 * in = n x width elements, out = n x (width/2) elements. */
int anemoi_jive_compress_batch(int field, int width, const uint64_t *in, uint64_t *out, size_t n, int device);

/* Jive::compress_k (anemoi_2_1/hasher.rs:105-110 accepts only k = 2; anemoi_4_3/hasher.rs:162-179
 * accepts k = 2, 4): out = n x (width/k) elements.  Other k -> ANEMOI_ERR_ARG. */
int anemoi_jive_compress_k_batch(int field, int width, int k, const uint64_t *in, uint64_t *out, size_t n,
                                 int device);

/* Sponge::merge of the 2-1 instances (anemoi_2_1/hasher.rs:87-92) = Jive compress of [left,right]:
 * pairs = n x 2 digests, out = n digests.  (The 4-3 merge is built on anemoi_permutation_batch by
 * the host shim so that the reference's behaviour there is preserved, see INTEGRATION.md.) */
int anemoi_merge_batch(int field, const uint64_t *pairs, uint64_t *out, size_t n, int device);

/* Sponge::hash_field (anemoi_2_1/hasher.rs:68-85, anemoi_4_3/hasher.rs:93-129) on n messages of
 * elems_per_msg elements each (contiguous); out = n digests (1 element each). */
int anemoi_hash_field_batch(int field, int width, const uint64_t *elems, size_t elems_per_msg, size_t n,
                            uint64_t *out, int device);

/* Sponge::hash (anemoi_2_1/hasher.rs:18-66, anemoi_4_3/hasher.rs:19-91) on n messages of msg_len
 * bytes each (contiguous): 31/47-byte little-endian chunks, 0x01 appended to a short last chunk. */
int anemoi_hash_bytes_batch(int field, int width, const uint8_t *msgs, size_t msg_len, size_t n, uint64_t *out,
                            int device);

/* Sponge::hash on n messages of DIFFERENT lengths: message i = bytes [offsets[i], offsets[i+1]) of `msgs`
 * (n + 1 non-decreasing byte offsets; an empty message hashes to the digest of the zero state, as the
 * reference's hash(b"") does).  Messages may come in any order: a wavefront costs what its longest message costs, so the
 * library stages them by descending block count and scatters the digests back (an unsorted long-tailed batch costs 1.02 x
 * the pre-sorted one). */
int anemoi_hash_bytes_ragged_batch(int field, int width, const uint8_t *msgs, const uint64_t *offsets, size_t n,
                                   uint64_t *out, int device);
/* Sponge::hash_field on n messages of DIFFERENT lengths: message i = elements [offsets[i], offsets[i+1]) of `elems`
 * (n + 1 non-decreasing offsets counted in ELEMENTS; an empty message hashes to the digest of the zero state).  The same
 * pipeline and the same bucketing by block count as the byte form. */
int anemoi_hash_field_ragged_batch(int field, int width, const uint64_t *elems, const uint64_t *offsets, size_t n,
                                   uint64_t *out, int device);

/* Root of the binary Merkle tree over 2^depth leaf digests built with the 2-1 instance's merge,
 * level by level (depth 0 returns the leaf; depth <= 30).  With ANEMOI_ALL_DEVICES each GPU
 * builds a contiguous subtree and the top log2(#GPUs) levels finish on the first device. */
int anemoi_merkle_root(int field, const uint64_t *leaves, unsigned depth, uint64_t *root, int device);

/* Merkle tree with every level retained (the realistic caller of merge):
 * tree = level 0 (the 2^depth leaves) | level 1 | ... | level depth (the root), 2^(depth+1) - 1
 * elements in all; node j of level l has children 2j, 2j+1 of level l-1. */
int anemoi_merkle_tree(int field, const uint64_t *leaves, unsigned depth, uint64_t *tree, int device);

/* Authentication path of leaf `index` out of a retained tree: `depth` sibling digests, bottom-up
 * (host-side indexing, no device work). */
int anemoi_merkle_path(int field, const uint64_t *tree, unsigned depth, size_t index, uint64_t *path);

/* Batched path verification on the GPU: item i recomputes the root from leaves[i], indices[i] and its
 * depth-element path (depth sequential merges per lane) and ok[i] = (it equals `root`, byte for byte: `root` must be a
 * reduced element -- leaves and paths may hold any pattern, see UNREDUCED INPUTS). */
int anemoi_merkle_verify_batch(int field, const uint64_t *leaves, const uint64_t *indices, const uint64_t *paths,
                               unsigned depth, size_t n, const uint64_t *root, uint8_t *ok, int device);

/* Arity-4 tree built with the 4-3 instance's Jive-4 compression (compress_k(.,4),
 * anemoi_4_3/hasher.rs:162-179): 4^depth4 leaf digests -> root (depth4 <= 15). */
int anemoi_merkle_root_arity4(int field, const uint64_t *leaves, unsigned depth4, uint64_t *root, int device);
/* The same tree with all levels retained: tree = level 0 (the 4^depth4 leaves) | level 1 | ... | root,
 * (4^(depth4+1) - 1) / 3 elements. */
int anemoi_merkle_tree_arity4(int field, const uint64_t *leaves, unsigned depth4, uint64_t *tree, int device);
/* Authentication path of leaf `index` out of such a tree: per level (bottom-up) the 3 siblings of the
 * node in child order, the node's own slot left out -- depth4 x 3 elements (host-side indexing only). */
int anemoi_merkle_path_arity4(int field, const uint64_t *tree, unsigned depth4, size_t index, uint64_t *path);
/* Batched verification on the GPU: item i rebuilds the root from leaves[i], indices[i] and its path
 * (depth4 Jive-4 compressions) and ok[i] = (it equals `root`). */
int anemoi_merkle_verify_arity4_batch(int field, const uint64_t *leaves, const uint64_t *indices,
                                      const uint64_t *paths, unsigned depth4, size_t n, const uint64_t *root,
                                      uint8_t *ok, int device);

/* canonical little-endian integers (< p) <-> Montgomery elements.  from_montgomery's output is
 * exactly AnemoiDigest::to_bytes (src/<f>/anemoi_x/digest.rs:42-46) when viewed as bytes. */
int anemoi_to_montgomery(int field, const uint64_t *in, uint64_t *out, size_t count, int device);
int anemoi_from_montgomery(int field, const uint64_t *in, uint64_t *out, size_t count, int device);

/* ---- instances given by their trait constants (src/traits.rs:36-76) -------------------------------
 * The reference's `Anemoi` trait is generic over NUM_COLUMNS, NUM_ROUNDS, ARK_C, ARK_D and an optional
 * MDS matrix, with hard-coded `mds_layer` arms for 1..6 columns and a matrix arm beyond
 * (src/traits.rs:136-304); the shipped instances use 1 and 2 columns only.  These entry points run any
 * such instance over one of the seven fields (ALPHA, BETA = generator, DELTA come with the field):
 * state width = 2 * num_columns elements, laid out [x_0 .. x_{c-1}, y_0 .. y_{c-1}] like the
 * reference's `state` slice.  All pointers are host memory; constants are Montgomery elements. */
typedef struct anemoi_generic_instance {
  int field;             /* field id */
  int num_columns;       /* NUM_COLUMNS, 1 .. ANEMOI_MAX_GENERIC_COLUMNS */
  int num_rounds;        /* NUM_ROUNDS, 1 .. 255 */
  const uint64_t *ark_c; /* ARK_C: num_rounds * num_columns elements (round-major, src/traits.rs:116-123) */
  const uint64_t *ark_d; /* ARK_D: same shape */
  const uint64_t *mds;   /* MDS: num_columns^2 elements, row-major; NULL selects the reference's hard-coded
                          * arm for num_columns <= 6 (and is an error beyond, like the reference's expect()) */
} anemoi_generic_instance;
#define ANEMOI_MAX_GENERIC_COLUMNS 16

/* The matrix the reference's hard-coded mds_layer arm for num_columns (1..6) applies to each half of the
 * state, as num_columns^2 row-major Montgomery elements (src/traits.rs:136-279, :307-323). */
int anemoi_generic_mds_matrix(int field, int num_columns, uint64_t *mds, int device);
/* Anemoi::permutation (src/traits.rs:370-378), in place: n states of 2 * num_columns elements */
int anemoi_generic_permutation_batch(const anemoi_generic_instance *inst, uint64_t *states, size_t n, int device);
/* Jive compress_k with the reference's argument rules (k even, k divides the state width;
 * anemoi_4_3/hasher.rs:162-179): n states -> n x (2 * num_columns / k) elements */
int anemoi_generic_jive_compress_k_batch(const anemoi_generic_instance *inst, int k, const uint64_t *in,
                                         uint64_t *out, size_t n, int device);
/* Sponge::hash_field / Sponge::hash with RATE_WIDTH = rate (1 .. 2 * num_columns - 1), digest = state[0]
 * (anemoi_4_3/hasher.rs:19-129): n messages of elems_per_msg elements / msg_len bytes -> n digests */
int anemoi_generic_hash_field_batch(const anemoi_generic_instance *inst, int rate, const uint64_t *elems,
                                    size_t elems_per_msg, size_t n, uint64_t *out, int device);
int anemoi_generic_hash_bytes_batch(const anemoi_generic_instance *inst, int rate, const uint8_t *msgs,
                                    size_t msg_len, size_t n, uint64_t *out, int device);
/* Element-wise x^ALPHA (exp_by_alpha, src/traits.rs:94-104; inverse = 0) or x^(1/ALPHA) (exp_by_inv_alpha,
 * src/<f>/sbox.rs; inverse = 1), in place -- the pair the reference's test_alpha checks against each other. */
int anemoi_exp_alpha_batch(int field, int inverse, uint64_t *elems, size_t n, int device);

/* ---- device-pointer API (buffers in the current device's HBM; asynchronous on `stream`) --- */
int anemoi_permutation_dev(int field, int width, void *d_states, size_t n, void *stream);
int anemoi_sbox_layer_dev(int field, int width, void *d_states, size_t n, void *stream);
int anemoi_jive_compress_k_dev(int field, int width, int k, const void *d_in, void *d_out, size_t n, void *stream);
int anemoi_hash_field_dev(int field, int width, const void *d_elems, size_t elems_per_msg, size_t n, void *d_out,
                          void *stream);
int anemoi_hash_bytes_dev(int field, int width, const void *d_msgs, size_t msg_len, size_t n, void *d_out,
                          void *stream);
/* d_offsets: n + 1 uint64 byte offsets into d_msgs (see anemoi_hash_bytes_ragged_batch).  Messages are processed IN THE
 * GIVEN ORDER, 64 (Anemoi-2-1) or 32 (4-3) consecutive messages per wavefront, and a wavefront runs as long as its longest
 * message: right for batches that are already grouped by length; for anything else use the bucketed form below.  Small
 * batches (up to the cut-offs of the equal-length sponge: options coop2d_max / coop2d43_max / coop_sponge_max) take the
 * latency kernels like an equal-length batch does -- 2 or 4 (Anemoi-2-1) / 1 or 2 (4-3) messages per wavefront.
 * The offsets are DEVICE memory, which no host-side check can read (the host forms answer ANEMOI_ERR_ARG for malformed
 * offsets before any device work).  What the device does with them: a pair that DECREASES (offsets[i+1] < offsets[i]) is
 * hashed as an EMPTY message -- its length is never the wrapped difference.  An offset beyond the end of d_msgs cannot be
 * recognised here, because this form is not told where d_msgs ends: that is undefined behaviour on the device (a memory
 * fault).  A caller that cannot vouch for its offsets uses the bucketed form, which is told and checks. */
int anemoi_hash_bytes_ragged_dev(int field, int width, const void *d_msgs, const void *d_offsets, size_t n, void *d_out,
                                 void *stream);
/* The same on an UNSORTED device-resident batch: the library first orders the messages by descending block count on the
 * device (a counting sort: zero, histogram, scan, placement -- four small launches on `stream`), runs the ragged kernels in
 * that order and writes every digest to its message's own index.  msgs_len: the bytes d_msgs holds.  d_scratch:
 * anemoi_ragged_scratch_bytes(n) bytes of device memory the caller owns (a status word, 65 536 counters, n 32-bit indices;
 * 4-byte aligned), contents undefined afterwards except for the status word; n < 2^32.  Only kernel launches on `stream`:
 * capturable like the other `_dev` functions (tests/test_gpu_capture.py replays it).
 * MALFORMED OFFSETS are reported, not followed: the histogram launch, which reads every offset pair anyway, sets bits in
 * the FIRST 32-BIT WORD OF d_scratch -- ANEMOI_RAGGED_DECREASING if some offsets[i+1] < offsets[i], ANEMOI_RAGGED_BEYOND_EXTENT
 * if offsets[n] > msgs_len -- and the sponge launch, seeing the word set, writes n all-zero digests and reads no byte of
 * d_msgs.  The function itself returns ANEMOI_OK (it only enqueues); the caller reads the word once the work on `stream`
 * has completed (hipMemcpyAsync of 4 bytes behind the call): 0 = every digest is valid (n = 0 enqueues nothing and leaves
 * the word as it is).  With non-decreasing offsets and
 * offsets[n] <= msgs_len every read lies inside d_msgs, so no device-resident value can make this call fault. */
#define ANEMOI_RAGGED_DECREASING 1
#define ANEMOI_RAGGED_BEYOND_EXTENT 2
size_t anemoi_ragged_scratch_bytes(size_t n);
int anemoi_hash_bytes_ragged_bucketed_dev(int field, int width, const void *d_msgs, size_t msgs_len, const void *d_offsets,
                                          size_t n, void *d_out, void *d_scratch, size_t scratch_bytes, void *stream);
/* The two calls above for messages of field elements (hash_field): d_elems = elements in ABI form (8-byte aligned),
 * d_offsets = n + 1 uint64 offsets counted in ELEMENTS, elems_len = the ELEMENTS d_elems holds.  Same kernels, same routing,
 * same scratch, same treatment of malformed offsets. */
int anemoi_hash_field_ragged_dev(int field, int width, const void *d_elems, const void *d_offsets, size_t n, void *d_out,
                                 void *stream);
int anemoi_hash_field_ragged_bucketed_dev(int field, int width, const void *d_elems, size_t elems_len, const void *d_offsets,
                                          size_t n, void *d_out, void *d_scratch, size_t scratch_bytes, void *stream);
/* d_scratch: at least 2^depth elements; d_root: 1 element; d_leaves is not modified. */
int anemoi_merkle_root_dev(int field, const void *d_leaves, unsigned depth, void *d_scratch, void *d_root,
                           void *stream);
/* d_tree: 2^(depth+1) - 1 elements, laid out as anemoi_merkle_tree; may alias d_leaves at its start. */
int anemoi_merkle_tree_dev(int field, const void *d_leaves, unsigned depth, void *d_tree, void *stream);
/* d_index: n x uint64; d_paths: n x depth elements; d_roots: n recomputed roots. */
int anemoi_merkle_climb_dev(int field, const void *d_leaves, const void *d_index, const void *d_paths,
                            unsigned depth, size_t n, void *d_roots, void *stream);
int anemoi_to_montgomery_dev(int field, const void *d_in, void *d_out, size_t count, void *stream);
int anemoi_from_montgomery_dev(int field, const void *d_in, void *d_out, size_t count, void *stream);


/* ---- run-time instances on device pointers (src/traits.rs:36-76, :136-304) ---------------------------
 * anemoi_generic_prepare() uploads an instance's constants to `device` ONCE and converts them into the
 * kernels' own form (the hard-coded arm's matrix included when mds == NULL); the handle then serves any
 * number of asynchronous calls on buffers in that device's HBM.  The caller's current device must be the
 * handle's device; the handle must outlive the work queued with it; anemoi_generic_destroy() frees it
 * (after the caller has synchronised its streams).  Errors and argument rules as for the _batch forms. */
typedef struct anemoi_generic_handle anemoi_generic_handle;
int anemoi_generic_prepare(const anemoi_generic_instance *inst, int device, anemoi_generic_handle **handle);
int anemoi_generic_destroy(anemoi_generic_handle *handle);
int anemoi_generic_permutation_dev(const anemoi_generic_handle *handle, void *d_states, size_t n, void *stream);
int anemoi_generic_jive_compress_k_dev(const anemoi_generic_handle *handle, int k, const void *d_in, void *d_out,
                                       size_t n, void *stream);
int anemoi_generic_hash_field_dev(const anemoi_generic_handle *handle, int rate, const void *d_elems,
                                  size_t elems_per_msg, size_t n, void *d_out, void *stream);
int anemoi_generic_hash_bytes_dev(const anemoi_generic_handle *handle, int rate, const void *d_msgs, size_t msg_len,
                                  size_t n, void *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ANEMOI_MI355X_H */
