// anemoi_kernels.h -- batch kernels and their launchers, templated on the field.
//
// one item per lane (throughput path):
//   k_permutation   Anemoi::permutation / sbox_layer   src/traits.rs:370-378, :326-358
//   k_jive          Jive::compress / compress_k        src/<f>/anemoi_2_1/hasher.rs:96-110, anemoi_4_3/hasher.rs:148-179
//   k_sponge        Sponge::hash / hash_field          src/<f>/anemoi_2_1/hasher.rs:18-85, anemoi_4_3/hasher.rs:19-129
//   k_merkle_climb  authentication-path verification   (depth x Sponge::merge, anemoi_2_1/hasher.rs:87-92)
//   k_mont_convert  canonical <-> Montgomery           (arkworks into_bigint / from_bigint)
// one Anemoi-4-3 state per lane PAIR (both columns' S-boxes run side by side):
//   k_permutation_pair, k_jive_pair, k_sponge_pair      (all width-4 entry points)
// a few items per wavefront, one limb per lane (latency path for small batches; anemoi_coop_kernels.h), on two arithmetics --
// LPR = 32: the two-row fold product of coop2d.h (an element on a pair of 16-lane DPP rows: the lowest latency),
// LPR = 16: the digit-serial scan of coop29.h (an element per row: twice the items per wavefront):
//   k_jive2_coop<F, LPR>       Jive::compress 2-1 / Sponge::merge: two (32) / four (16) items per wavefront;
//                              <F, 64> = one item per wavefront (four-row fold / rounds 1-2's scan), kept for A/B
//   k_jive4_coop<F, K, LPR>    Jive::compress(_k) 4-3: one state per wavefront, a column per row pair (32) / two, a column
//                              per row (16)
//   k_sponge_coop              Sponge::hash / hash_field on small batches of equal-length messages (both widths, both LPR)
//   k_permutation_coop, k_merkle_climb_coop   Anemoi::permutation / path verification on small batches
// instances given by run-time trait constants, one state per NUM_COLUMNS lanes (anemoi_generic.h):
//   k_permutation_cols, k_jive_cols, k_sponge_cols, and the element-wise k_exp_alpha
//
// Data layout in HBM: array-of-states, each state `W` elements of N 32-bit limbs (= the reference's
// `&[Felt]` bytes).  A workgroup's states are contiguous, so it moves them with 16-byte-per-lane
// coalesced loads/stores and transposes through LDS into one-state-per-lane registers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "anemoi_perm.h"
#ifndef ANEMOI_BOUNDS_WALK   // (the host walk brings its own recording arithmetic for the latency kernels)
#include "coop29.h"
#include "coop2d.h"
#endif
#include "options.h"

namespace anemoi {

constexpr int kBlock = 64;  // one wavefront per workgroup: the LDS window table is lane-private

// Register budget: ANEMOI_WAVES waves per SIMD (0 = let the compiler choose; the shipped setting).
// Forcing 5-8 waves/SIMD was measured and is slower: the kernels are VALU-issue bound at 3 waves
// (profiles/r01/ab_occupancy_variants.txt).
#if ANEMOI_WAVES > 0   // (A/B builds only, build_config.h)
#define ANEMOI_KERNEL __global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(ANEMOI_WAVES, 8)))
#else
#define ANEMOI_KERNEL __global__ __launch_bounds__(kBlock)
#endif

// build_config.h: ANEMOI_HOLD_INPUTS_MAX_NL = 13 (k_jive 2-1 keeps the feed-forward sum in VGPRs up to this many limbs),
// ANEMOI_WIN = 3 (3 LDS entries per lane -> 13 waves per CU; LABNOTES.md section 3.3)
template <int N>
struct KernelCfg {
  static constexpr int WIN = ANEMOI_WIN;  // sliding-window bits -> 2^(WIN-1) odd powers per lane in LDS
};

// (ANEMOI_LDS_ENTRIES_9, A/B builds: LDS table entries requested by the 9-limb fields; 0 = what the window needs)
template <class A, int WIN>
constexpr size_t lds_table_bytes() {
  // x^3, x^5, ..: x itself stays in VGPRs
  constexpr int need = (1 << (WIN - 1)) - 1;
  constexpr int entries = (A::NL < 13 && ANEMOI_LDS_ENTRIES_9 > need) ? ANEMOI_LDS_ENTRIES_9 : need;
  return size_t(entries) * A::NQ * 16 * kBlock;
}

template <class A, int WIN, int W>
constexpr size_t lds_bytes() {
  size_t tab = lds_table_bytes<A, WIN>(), stage = size_t(W) * A::NABI * 4 * kBlock;
  return tab > stage ? tab : stage;
}

// ---- coalesced block I/O through LDS -------------------------------------------------------------
// The block owns items [blk0, blk0 + cnt) of `per_item` uint4 each, contiguous in global memory.
template <int PER_ITEM>
__device__ __forceinline__ void block_load(uint4* lds, const uint4* __restrict__ g, size_t blk0, int cnt) {
  const uint4* src = g + blk0 * PER_ITEM;
  const int total = cnt * PER_ITEM;
#pragma unroll
  for (int i = 0; i < PER_ITEM; i++) {
    int idx = i * kBlock + threadIdx.x;
    if (idx < total) lds[idx] = src[idx];
  }
  __syncthreads();
}

template <int PER_ITEM>
__device__ __forceinline__ void block_store(uint4* lds, uint4* __restrict__ g, size_t blk0, int cnt) {
  __syncthreads();
  uint4* dst = g + blk0 * PER_ITEM;
  const int total = cnt * PER_ITEM;
#pragma unroll
  for (int i = 0; i < PER_ITEM; i++) {
    int idx = i * kBlock + threadIdx.x;
    if (idx < total) dst[idx] = lds[idx];
  }
}

// ABI element `slot` of the staging area -> registers (internal form), and back
template <class A>
__device__ __forceinline__ void lds_get(const uint4* lds, int slot, typename A::Fe& v) {
  uint32_t w[A::NABI];
#pragma unroll
  for (int q = 0; q < A::NABI / 4; q++) {
    uint4 t = lds[slot * (A::NABI / 4) + q];
    w[4 * q] = t.x;
    w[4 * q + 1] = t.y;
    w[4 * q + 2] = t.z;
    w[4 * q + 3] = t.w;
  }
  A::from_abi(v, w);
}

template <class A>
__device__ __forceinline__ void lds_put(uint4* lds, int slot, const typename A::Fe& v) {
  uint32_t w[A::NABI];
  A::to_abi(w, v);
#pragma unroll
  for (int q = 0; q < A::NABI / 4; q++)
    lds[slot * (A::NABI / 4) + q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}

template <class A>
__device__ __forceinline__ LdsTable<A> make_table(uint4* lds) {
  LdsTable<A> t;
  t.base = lds + threadIdx.x;
  t.stride = kBlock;
  return t;
}

// ---- kernels -------------------------------------------------------------------------------------

// SBOX_ONLY: apply just Anemoi::sbox_layer (src/traits.rs:326-358) -- the unit the reference's
// test_sbox KATs pin (src/<f>/anemoi_x/mod.rs:68).
template <int FIELD, int W, bool SBOX_ONLY>
ANEMOI_KERNEL void k_permutation(uint4* __restrict__ states, size_t n, PermConsts pc) {
  using F = FieldC<FIELD>;
  using A = ArithFor<FIELD>;
  constexpr int WIN = KernelCfg<F::N>::WIN, PER = W * A::NABI / 4;
  extern __shared__ uint4 lds[];
  const size_t blk0 = size_t(blockIdx.x) * kBlock;
  const int cnt = n - blk0 < size_t(kBlock) ? int(n - blk0) : kBlock;
  block_load<PER>(lds, states, blk0, cnt);
  typename A::Fe st[W];
  static_for<0, W>([&](auto i) { lds_get<A>(lds, threadIdx.x * W + i, st[i]); });
  __syncthreads();
  if (SBOX_ONLY) sbox_layer<F, A, W, WIN>(st, pc, make_table<A>(lds));
  else permutation<F, A, W, WIN>(st, pc, make_table<A>(lds));
  __syncthreads();
  static_for<0, W>([&](auto i) { lds_put<A>(lds, threadIdx.x * W + i, st[i]); });
  block_store<PER>(lds, states, blk0, cnt);
}

// out[i] = sum_{j<k} in[i + c*j] + perm(in)[i + c*j], c = W/k  (k = 2: c = W/2 outputs; k = 4: 1 output)
// One block of 64 states (the work of one workgroup = one wavefront).
template <int FIELD, int W, int K>
__device__ __forceinline__ void jive_block(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n, const PermConsts& pc,
                                           size_t block, uint4* lds) {
  using F = FieldC<FIELD>;
  using A = ArithFor<FIELD>;
  constexpr int WIN = KernelCfg<F::N>::WIN, PER = W * A::NABI / 4, C = W / K;
  const size_t blk0 = block * kBlock;
  const int cnt = n - blk0 < size_t(kBlock) ? int(n - blk0) : kBlock;
  block_load<PER>(lds, in, blk0, cnt);
  typename A::Fe st[W], sum[C];
  static_for<0, W>([&](auto i) { lds_get<A>(lds, threadIdx.x * W + i, st[i]); });
  __syncthreads();
  // Jive feed-forward sum of the inputs.  W = 2: one element, kept in VGPRs across the permutation
  // (+9.6 % on Jubjub against re-reading; on the 13-limb layout 138 instead of 130 VGPRs, same speed,
  // and the HBM traffic drops from 1.43x to the algorithmic bytes).  The W = 4 one-state-per-lane kernel
  // would spill, so it fetches the inputs again at the end (from L2 / Infinity Cache).
  constexpr bool kHoldInputs = W == 2 && A::NL <= ANEMOI_HOLD_INPUTS_MAX_NL;
  typename A::Fe insum[kHoldInputs ? C : 1];
  if constexpr (kHoldInputs) {
#pragma unroll
    for (int i = 0; i < C; i++) {
      insum[i] = st[i];
#pragma unroll
      for (int j = 1; j < K; j++) A::add(insum[i], insum[i], st[i + C * j]);
    }
  }
  permutation<F, A, W, WIN>(st, pc, make_table<A>(lds));
#pragma unroll
  for (int i = 0; i < C; i++) {
    sum[i] = st[i];
#pragma unroll
    for (int j = 1; j < K; j++) A::add(sum[i], sum[i], st[i + C * j]);
  }
  if constexpr (kHoldInputs) {
#pragma unroll
    for (int i = 0; i < C; i++) A::add(sum[i], sum[i], insum[i]);
  } else {
    __syncthreads();
    block_load<PER>(lds, in, blk0, cnt);
    static_for<0, W>([&](auto i) { lds_get<A>(lds, threadIdx.x * W + i, st[i]); });
#pragma unroll
    for (int i = 0; i < C; i++) {
#pragma unroll
      for (int j = 0; j < K; j++) A::add(sum[i], sum[i], st[i + C * j]);
    }
  }
  __syncthreads();
  static_for<0, C>([&](auto i) { lds_put<A>(lds, threadIdx.x * C + i, sum[i]); });
  block_store<C * A::NABI / 4>(lds, out, blk0, cnt);
}

template <int FIELD, int W, int K>
ANEMOI_KERNEL void k_jive(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n,
                                                 PermConsts pc) {
  extern __shared__ uint4 lds[];
  jive_block<FIELD, W, K>(in, out, n, pc, blockIdx.x, lds);
}

#if ANEMOI_AB_BUILD
// RECORDED NEGATIVE (laboratory builds only; tools/exp_jive_queue.py, profiles/r05/jive_work_queue_not_adopted.txt): the
// same, the blocks handed out by a counter -- a workgroup takes the next block when it has finished one.  The thought:
// the sampled clocks of the XCDs differ by 2-5 % under this load and kernel time x the SLOWEST sampled clock is the same
// 228.4 Mcycles on every box, as if the launch waited for the slowest XCD to work off its statically dealt eighth; with a
// queue the faster XCDs would take more.  Measured in one process: 100.00 against 100.24 ms at 2^20, 200.56 against
// 200.27 at 2^21 -- nothing.  Whatever makes the slowest sampled clock the right one, it is not static dealing.
// `queue[0]` must be 0 at launch; grid = the resident workgroups of the device.
// XCD ACCOUNTING (round 6; tools/exp_xcd_accounting.py, profiles/r06/xcd_accounting.txt): what each of the 8 XCDs did in
// one launch -- blocks worked, the wall clock (100 MHz) of its first block's start and of its last block's end, and the
// wall ticks / shader cycles its blocks took in sum (their ratio is the clock the XCD held UNDER ITS OWN WORK, measured by
// the work's own wavefronts; cycles per block says whether a block costs the same everywhere).  Five atomics per block of
// ~19 ms.  `acct` (may be null): 8 records; the caller zeroes them and sets first_start to ~0.
struct XcdAcct {
  unsigned long long blocks, first_start, last_end, wall_ticks, cycles;
};
struct AcctScope {   // (the start stamps wait in LDS: four more live SGPRs spilled k_jive_queue<bls12_381> to 169 VGPRs = 2 waves per SIMD)
  XcdAcct* a;
  __device__ __forceinline__ static unsigned long long* stamps() {
    __shared__ unsigned long long st[2];
    return st;
  }
  __device__ __forceinline__ explicit AcctScope(XcdAcct* acct) : a(acct) {
    if (a && threadIdx.x == 0) {
      unsigned long long* st = stamps();
      st[0] = __builtin_amdgcn_s_memrealtime(), st[1] = __builtin_amdgcn_s_memtime();
    }
  }
  __device__ __forceinline__ void end() {
    if (!a || threadIdx.x) return;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long* st = stamps();
    const unsigned long long t0 = st[0], c0 = st[1];
    XcdAcct& x = a[__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) & 7u];   // XCC_ID (as k_clock_sampler reads it)
    atomicAdd(&x.blocks, 1ull);
    atomicMin(&x.first_start, t0);
    atomicMax(&x.last_end, t1);
    atomicAdd(&x.wall_ticks, t1 - t0);
    atomicAdd(&x.cycles, c1 - c0);
  }
};
// the shipped kernel's dealing (block = blockIdx.x) with the accounting around it
template <int FIELD, int W, int K>
ANEMOI_KERNEL void k_jive_acct(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n, PermConsts pc, XcdAcct* acct) {
  extern __shared__ uint4 lds[];
  AcctScope sc(acct);
  jive_block<FIELD, W, K>(in, out, n, pc, blockIdx.x, lds);
  sc.end();
}
template <int FIELD, int W, int K>
ANEMOI_KERNEL void k_jive_queue(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n, PermConsts pc,
                                uint32_t* __restrict__ queue, XcdAcct* acct) {
  extern __shared__ uint4 lds[];
  const uint32_t nblocks = uint32_t((n + kBlock - 1) / kBlock);
  for (;;) {
    uint32_t b = 0;
    if (threadIdx.x == 0) b = atomicAdd(queue, 1u);
    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    if (b >= nblocks) break;
    AcctScope sc(acct);
    jive_block<FIELD, W, K>(in, out, n, pc, b, lds);
    sc.end();
    __syncthreads();   // the staging area is reused by the next block
  }
}
// EXPERIMENT: one block per workgroup as in k_jive, but WHICH block is a ticket drawn from a counter, and the grid holds
// more workgroups than there are blocks: the hardware still deals the workgroups to the XCDs round-robin, a faster XCD
// gets through its share sooner and so draws more tickets, the surplus workgroups of the slower ones find none and leave.
template <int FIELD, int W, int K>
ANEMOI_KERNEL void k_jive_ticket(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n, PermConsts pc,
                                 uint32_t* __restrict__ counter, XcdAcct* acct) {
  extern __shared__ uint4 lds[];
  uint32_t b = 0;
  if (threadIdx.x == 0) b = atomicAdd(counter, 1u);
  b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
  if (b >= uint32_t((n + kBlock - 1) / kBlock)) return;
  AcctScope sc(acct);
  jive_block<FIELD, W, K>(in, out, n, pc, b, lds);
  sc.end();
}
#endif  // ANEMOI_AB_BUILD

// One chunk of a byte message -> internal Montgomery element (from_le_bytes_mod_order + the
// reference's padding rule: a SHORT last chunk gets a 0x01 byte appended; hasher.rs:36-57).
template <class F, class A>
__device__ __forceinline__ void chunk_to_fe(typename A::Fe& e, const uint8_t* __restrict__ p, int len) {
  uint32_t w[A::NABI];
#pragma unroll
  for (int i = 0; i < A::NABI; i++) {
    uint32_t v = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int pos = 4 * i + b;
      uint32_t byte = 0;
      if (pos < F::kChunk) {
        if (pos < len) byte = p[pos];
        else if (pos == len) byte = 1;  // only reachable when len < kChunk
      }
      v |= byte << (8 * b);
    }
    w[i] = v;
  }
  // value < 2^(8*kChunk+1) < p: already reduced; enter Montgomery form
  A::from_int(e, w);
}

// One launch of a sponge kernel absorbs one SEGMENT of every message: elements [e0, e0 + seg_elems) (to the
// end, padding included, when `last`).  The whole-message launch is the segment with first = last = 1.
// Long messages from host memory are fed segment by segment (capi.hip: the copy of segment c + 1 runs under
// the kernel of segment c), the sponge state carried between launches in `state` as ABI elements.
struct SpongeSeg {
  uint32_t* state;   // [n][W] ABI elements, read unless `first`, written unless `last`; may be null if first && last
  size_t e0;         // index (within the message) of the segment's first element: a multiple of RATE
  size_t total_len;  // length of the WHOLE message: bytes (BYTES) or elements
  int first, last;
};

// Element e of a message as the sponge absorbs it: chunk e of a byte message (BYTES) or ABI element e,
// and the padding element 1 once the message is exhausted (e == num, only when num % RATE != 0).
// `seg` points at the segment's first element; total_len is the whole message's length.
template <class F, class A, bool BYTES>
__device__ __forceinline__ void sponge_element(typename A::Fe& el, const uint8_t* __restrict__ seg, size_t e, size_t e0,
                                               size_t num, size_t total_len) {
  if (e < num) {
    if (BYTES) {
      const size_t left = total_len - e * F::kChunk;
      chunk_to_fe<F, A>(el, seg + (e - e0) * F::kChunk, left < size_t(F::kChunk) ? int(left) : F::kChunk);
    } else {
      const uint32_t* src32 = (const uint32_t*)seg + (e - e0) * A::NABI;
      uint32_t w[A::NABI];
#pragma unroll
      for (int l = 0; l < A::NABI; l++) w[l] = src32[l];
      A::from_abi(el, w);
    }
  } else {
    A::set_one(el);
  }
}

// sponge state element <-> the carry buffer (ABI words, canonical)
template <class A>
__device__ __forceinline__ void state_load(typename A::Fe& v, const uint32_t* __restrict__ p) {
  uint32_t w[A::NABI];
#pragma unroll
  for (int l = 0; l < A::NABI; l++) w[l] = p[l];
  A::from_abi(v, w);
}
template <class A>
__device__ __forceinline__ void state_store(uint32_t* __restrict__ p, const typename A::Fe& v) {
  uint32_t w[A::NABI];
  A::to_abi(w, v);
#pragma unroll
  for (int l = 0; l < A::NABI; l++) p[l] = w[l];
}

// Sponge over `num` elements per message (BYTES: taken from msg_len-byte messages; else ABI
// elements).  Unified rule (== both hasher.rs variants): absorb into state[i]; permute when
// i == RATE; if num % RATE != 0 absorb a final 1 and permute; digest = state[0].
template <int FIELD, int W, bool BYTES>
ANEMOI_KERNEL void k_sponge(const void* __restrict__ src, size_t per_msg, size_t n,
                                                   uint4* __restrict__ out, PermConsts pc, SpongeSeg seg) {
  using F = FieldC<FIELD>;
  using A = ArithFor<FIELD>;
  constexpr int WIN = KernelCfg<F::N>::WIN, RATE = W - 1;
  extern __shared__ uint4 lds[];
  const size_t blk0 = size_t(blockIdx.x) * kBlock;
  const int cnt = n - blk0 < size_t(kBlock) ? int(n - blk0) : kBlock;
  const size_t item = blk0 + (threadIdx.x < cnt ? threadIdx.x : 0);  // idle lanes redo item blk0
  // per_msg = this launch's bytes / elements per message (the stride of `src`); seg.total_len = the whole message
  const size_t num = BYTES ? (seg.total_len + F::kChunk - 1) / F::kChunk : seg.total_len;
  const size_t total = num + (num % RATE == 0 ? 0 : 1);
  const size_t e_end = seg.last ? total : seg.e0 + (BYTES ? per_msg / F::kChunk : per_msg);
  const uint8_t* msg = (const uint8_t*)src + (BYTES ? item * per_msg : item * per_msg * A::NABI * 4);
  typename A::Fe st[W];
  if (seg.first) static_for<0, W>([&](auto i) { A::set_zero(st[i]); });
  else static_for<0, W>([&](auto i) { state_load<A>(st[i], seg.state + (item * W + i) * A::NABI); });
  int pos = 0;
  const LdsTable<A> tab = make_table<A>(lds);
#pragma nounroll
  for (size_t e = seg.e0; e < e_end; e++) {
    typename A::Fe el;
    sponge_element<F, A, BYTES>(el, msg, e, seg.e0, num, seg.total_len);
    // pos is wave-uniform (every message has the same length)
    if (RATE == 1 || pos == 0) A::add(st[0], st[0], el);
    else if (pos == 1) A::add(st[1], st[1], el);
    else A::add(st[RATE - 1], st[RATE - 1], el);
    pos++;
    if (pos == RATE || e == total - 1) {
      permutation<F, A, W, WIN>(st, pc, tab);
      pos = 0;
    }
  }
  if (!seg.last) {  // carry the state to the next segment's launch
    if (threadIdx.x < cnt) static_for<0, W>([&](auto i) { state_store<A>(seg.state + (item * W + i) * A::NABI, st[i]); });
    return;
  }
  __syncthreads();
  lds_put<A>(lds, threadIdx.x, st[0]);
  block_store<A::NABI / 4>(lds, out, blk0, cnt);
}

// ---- Anemoi-4-3 kernels with one state per lane PAIR (anemoi_perm.h, "two columns on two lanes") -------
// A workgroup of 64 lanes owns 32 states; lane t works on state t/2, column t%2.
constexpr int kPairStates = kBlock / 2;

template <int FIELD, bool SBOX_ONLY>
ANEMOI_KERNEL void k_permutation_pair(uint4* __restrict__ states, size_t n, PermConsts pc) {
  using F = FieldC<FIELD>;
  using A = ArithFor<FIELD>;
  constexpr int WIN = KernelCfg<F::N>::WIN, PER = 2 * A::NABI / 4;  // uint4 per lane
  extern __shared__ uint4 lds[];
  const size_t st0 = size_t(blockIdx.x) * kPairStates;
  const int cnt = n - st0 < size_t(kPairStates) ? int(n - st0) : kPairStates;
  const bool odd = threadIdx.x & 1;
  const int s = threadIdx.x >> 1;
  block_load<PER>(lds, states, size_t(blockIdx.x) * kBlock, 2 * cnt);
  typename A::Fe x, y;
  lds_get<A>(lds, s * 4 + (odd ? 1 : 0), x);
  lds_get<A>(lds, s * 4 + 2 + (odd ? 1 : 0), y);
  __syncthreads();
  if (SBOX_ONLY) flystel<F, A, WIN>(x, y, pc, make_table<A>(lds));
  else permutation_pair<F, A, WIN>(x, y, odd, pc, make_table<A>(lds));
  __syncthreads();
  lds_put<A>(lds, s * 4 + (odd ? 1 : 0), x);
  lds_put<A>(lds, s * 4 + 2 + (odd ? 1 : 0), y);
  block_store<PER>(lds, states, size_t(blockIdx.x) * kBlock, 2 * cnt);
}

// Jive on Anemoi-4-3 (anemoi_4_3/hasher.rs:148-179).  K = 2: out[i] = e_i + e_{i+2} + s_i + s_{i+2}, i.e.
// lane-local (x_in + y_in + x_out + y_out); K = 4: the two lanes' sums are added.
template <int FIELD, int K>
ANEMOI_KERNEL void k_jive_pair(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n, PermConsts pc) {
  using F = FieldC<FIELD>;
  using A = ArithFor<FIELD>;
  constexpr int WIN = KernelCfg<F::N>::WIN, PER = 2 * A::NABI / 4;
  extern __shared__ uint4 lds[];
  const size_t st0 = size_t(blockIdx.x) * kPairStates;
  const int cnt = n - st0 < size_t(kPairStates) ? int(n - st0) : kPairStates;
  const bool odd = threadIdx.x & 1;
  const int s = threadIdx.x >> 1;
  block_load<PER>(lds, in, size_t(blockIdx.x) * kBlock, 2 * cnt);
  typename A::Fe x, y, sum, t;
  lds_get<A>(lds, s * 4 + (odd ? 1 : 0), x);
  lds_get<A>(lds, s * 4 + 2 + (odd ? 1 : 0), y);
  __syncthreads();
  A::add(sum, x, y);  // feed-forward part, one element per lane, kept in registers
  permutation_pair<F, A, WIN>(x, y, odd, pc, make_table<A>(lds));
  A::add(t, x, y);
  A::add(sum, sum, t);
  __syncthreads();
  if (K == 2) {
    lds_put<A>(lds, s * 2 + (odd ? 1 : 0), sum);
    block_store<A::NABI / 4>(lds, out, size_t(blockIdx.x) * kBlock, 2 * cnt);
  } else {
    fe_exchange<A>(t, sum);
    A::add(sum, sum, t);
    if (A::kLoose) A::settle(sum);
    // only the even lane's copy is stored (both lanes hold the same total)
    typename A::Fe dummy = sum;
    if (!odd) lds_put<A>(lds, s, dummy);
    else lds_put<A>(lds, kPairStates + s, dummy);  // scratch slots beyond the 32 outputs
    block_store<A::NABI / 4>(lds, out, st0, cnt);
  }
}

// Sponge on Anemoi-4-3 (anemoi_4_3/hasher.rs:19-129), one message per lane pair: both lanes decode
// the same element, the lane that owns state[pos] absorbs it (state[0], state[1] = x of the even / odd
// lane, state[2] = y of the even lane).
template <int FIELD, bool BYTES>
ANEMOI_KERNEL void k_sponge_pair(const void* __restrict__ src, size_t per_msg, size_t n, uint4* __restrict__ out,
                                 PermConsts pc, SpongeSeg seg) {
  using F = FieldC<FIELD>;
  using A = ArithFor<FIELD>;
  constexpr int WIN = KernelCfg<F::N>::WIN, RATE = 3;
  extern __shared__ uint4 lds[];
  const size_t st0 = size_t(blockIdx.x) * kPairStates;
  const int cnt = n - st0 < size_t(kPairStates) ? int(n - st0) : kPairStates;
  const bool odd = threadIdx.x & 1;
  const int s = threadIdx.x >> 1;
  const size_t item = st0 + (s < cnt ? s : 0);  // idle lane pairs redo item st0
  const size_t num = BYTES ? (seg.total_len + F::kChunk - 1) / F::kChunk : seg.total_len;
  const size_t total = num + (num % RATE == 0 ? 0 : 1);
  const size_t e_end = seg.last ? total : seg.e0 + (BYTES ? per_msg / F::kChunk : per_msg);
  const uint8_t* msg = (const uint8_t*)src + (BYTES ? item * per_msg : item * per_msg * A::NABI * 4);
  typename A::Fe x, y;
  if (seg.first) {
    A::set_zero(x);
    A::set_zero(y);
  } else {  // state[0], state[1] = x of the even / odd lane; state[2], state[3] = y of the even / odd lane
    state_load<A>(x, seg.state + (item * 4 + (odd ? 1 : 0)) * A::NABI);
    state_load<A>(y, seg.state + (item * 4 + 2 + (odd ? 1 : 0)) * A::NABI);
  }
  int pos = 0;
  const LdsTable<A> tab = make_table<A>(lds);
#pragma nounroll
  for (size_t e = seg.e0; e < e_end; e++) {
    typename A::Fe el, t;
    sponge_element<F, A, BYTES>(el, msg, e, seg.e0, num, seg.total_len);
    // pos is wave-uniform: state[0] -> even.x, state[1] -> odd.x, state[2] -> even.y
    if (pos < 2) {
      A::add(t, x, el);
      fe_select<A>(x, odd == (pos == 1), t, x);
    } else {
      A::add(t, y, el);
      fe_select<A>(y, !odd, t, y);
    }
    pos++;
    if (pos == RATE || e == total - 1) {
      permutation_pair<F, A, WIN>(x, y, odd, pc, tab);
      pos = 0;
    }
  }
  if (!seg.last) {
    if (s < cnt) {
      state_store<A>(seg.state + (item * 4 + (odd ? 1 : 0)) * A::NABI, x);
      state_store<A>(seg.state + (item * 4 + 2 + (odd ? 1 : 0)) * A::NABI, y);
    }
    return;
  }
  __syncthreads();
  // digest = state[0] = the even lane's x; odd lanes write to scratch slots beyond the outputs
  lds_put<A>(lds, odd ? kPairStates + s : s, x);
  block_store<A::NABI / 4>(lds, out, st0, cnt);
}

// ---- sponge over messages of DIFFERENT lengths (Sponge::hash on a ragged batch) ----------------------------
// Message i = bytes [off[i], off[i+1]) of `msgs`.  Every lane walks its own message; the wavefront runs as many
// rate-blocks as its longest message needs.  A rate-block is: absorb up to RATE elements (a message's last
// block may be short: the cells it does not reach get nothing, which is what the reference's
// "permute when i == RATE or at the last element" does), then permute.  The permutation is executed by all
// lanes in every block (wave-uniform control flow around the big inlined body; the lane pairs of the 4-3
// kernel need each other in the linear layer); a lane whose message has ended latches its digest after its
// own last block and lets its state run on unobserved.  Callers that care about throughput sort the batch
// by length: a wavefront costs what its longest message costs.
template <class T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int d = 32; d; d >>= 1) {
    const T o = __shfl_xor(v, d);
    v = o > v ? o : v;
  }
  return v;
}

// digest -> out[message], straight from the lane (the bucketed order scatters its outputs)
template <class A>
__device__ __forceinline__ void store_digest(uint4* __restrict__ dst, const typename A::Fe& v) {
  uint32_t w[A::NABI];
  A::to_abi(w, v);
#pragma unroll
  for (int q = 0; q < A::NABI / 4; q++) dst[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}

// `order` (may be null): slot j of the launch works on message order[j] and its digest goes to out[order[j]] -- the
// device-side bucketing of anemoi_hash_bytes_ragged_bucketed_dev (k_ragged_hist / _scan / _place below): messages by
// descending block count, so that the lanes of a wavefront run out of blocks together.
// BYTES: messages of bytes, offsets in bytes (Sponge::hash); else messages of ABI elements, offsets in ELEMENTS (hash_field).
// MALFORMED OFFSETS (they are device memory: no host check can see them).  A decreasing pair is read as an EMPTY message
// (ragged_len), never as a length that wrapped.  `status` (may be null): the word the bucketing kernels leave -- non-zero
// when a pair decreases or the last offset lies beyond the blob; the launch then writes the empty message's digest (zero
// words) for every message and reads no message byte.  Block counts stay 64-bit: a message of >= 2^32 blocks (>= 133 GB)
// fits this GPU's memory, a narrowing would hash a prefix of it.
__device__ __forceinline__ uint64_t ragged_len(const uint64_t* __restrict__ off, size_t item) {
  const uint64_t o0 = off[item], o1 = off[item + 1];
  return o1 >= o0 ? o1 - o0 : 0;
}
__device__ __forceinline__ bool ragged_refused(const uint32_t* __restrict__ status) { return status && *status != 0; }
template <int Q>
__device__ __forceinline__ void store_zero_digest(uint4* __restrict__ dst) {
#pragma unroll
  for (int q = 0; q < Q; q++) dst[q] = make_uint4(0, 0, 0, 0);
}
template <int FIELD, bool BYTES>
ANEMOI_KERNEL void k_sponge_ragged(const uint8_t* __restrict__ msgs, const uint64_t* __restrict__ off, size_t n,
                                   uint4* __restrict__ out, PermConsts pc, const uint32_t* __restrict__ order,
                                   const uint32_t* __restrict__ status) {
  using F = FieldC<FIELD>;
  using A = ArithFor<FIELD>;
  constexpr int WIN = KernelCfg<F::N>::WIN;  // Anemoi-2-1: RATE = 1, a block is one element
  extern __shared__ uint4 lds[];
  const size_t blk0 = size_t(blockIdx.x) * kBlock;
  const int cnt = n - blk0 < size_t(kBlock) ? int(n - blk0) : kBlock;
  const size_t slot = blk0 + (threadIdx.x < cnt ? threadIdx.x : 0);  // idle lanes redo item blk0
  const size_t item = order ? size_t(order[slot]) : slot;
  if (ragged_refused(status)) {  // wave-uniform
    if (threadIdx.x < cnt) store_zero_digest<A::NABI / 4>(out + item * (A::NABI / 4));
    return;
  }
  const uint64_t o0 = off[item], len = ragged_len(off, item);
  const uint8_t* msg = msgs + o0 * (BYTES ? 1 : A::NABI * 4);
  const uint64_t num = BYTES ? len / F::kChunk + (len % F::kChunk ? 1 : 0) : len;  // RATE = 1: no padding element ever
  const uint64_t blocks = num, bmax = wave_max(blocks);
  typename A::Fe st[2], dig;
  A::set_zero(st[0]);
  A::set_zero(st[1]);
  A::set_zero(dig);  // the empty message hashes to state[0] of the zero state
  const LdsTable<A> tab = make_table<A>(lds);
#pragma nounroll
  for (uint64_t b = 0; b < bmax; b++) {
    if (b < blocks) {
      typename A::Fe el;
      sponge_element<F, A, BYTES>(el, msg, b, 0, num, len);
      A::add(st[0], st[0], el);
    }
    permutation<F, A, 2, WIN>(st, pc, tab);
    fe_select<A>(dig, b + 1 == blocks, st[0], dig);
  }
  if (order) {
    if (threadIdx.x < cnt) store_digest<A>(out + item * (A::NABI / 4), dig);
    return;
  }
  __syncthreads();
  lds_put<A>(lds, threadIdx.x, dig);
  block_store<A::NABI / 4>(lds, out, blk0, cnt);
}

template <int FIELD, bool BYTES>
ANEMOI_KERNEL void k_sponge_ragged_pair(const uint8_t* __restrict__ msgs, const uint64_t* __restrict__ off, size_t n,
                                        uint4* __restrict__ out, PermConsts pc, const uint32_t* __restrict__ order,
                                        const uint32_t* __restrict__ status) {
  using F = FieldC<FIELD>;
  using A = ArithFor<FIELD>;
  constexpr int WIN = KernelCfg<F::N>::WIN, RATE = 3;
  extern __shared__ uint4 lds[];
  const size_t st0 = size_t(blockIdx.x) * kPairStates;
  const int cnt = n - st0 < size_t(kPairStates) ? int(n - st0) : kPairStates;
  const bool odd = threadIdx.x & 1;
  const int s = threadIdx.x >> 1;
  const size_t slot = st0 + (s < cnt ? s : 0);  // idle lane pairs redo item st0
  const size_t item = order ? size_t(order[slot]) : slot;
  if (ragged_refused(status)) {  // wave-uniform
    if (!odd && s < cnt) store_zero_digest<A::NABI / 4>(out + item * (A::NABI / 4));
    return;
  }
  const uint64_t o0 = off[item], len = ragged_len(off, item);
  const uint8_t* msg = msgs + o0 * (BYTES ? 1 : A::NABI * 4);
  const uint64_t num = BYTES ? len / F::kChunk + (len % F::kChunk ? 1 : 0) : len;
  const uint64_t total = num + (num % RATE == 0 ? 0 : 1);  // + the padding element 1 (num <= 2^64 - 3 then: no wrap)
  const uint64_t blocks = total / RATE + (total % RATE ? 1 : 0), bmax = wave_max(blocks);
  typename A::Fe x, y, dig;
  A::set_zero(x);
  A::set_zero(y);
  A::set_zero(dig);
  const LdsTable<A> tab = make_table<A>(lds);
#pragma nounroll
  for (uint64_t b = 0; b < bmax; b++) {
    // state[0] -> even.x, state[1] -> odd.x, state[2] -> even.y; both lanes of a pair see the same message
    static_for<0, RATE>([&](auto r) {
      const size_t e = size_t(b) * RATE + r;
      if (e < total) {
        typename A::Fe el, t;
        sponge_element<F, A, BYTES>(el, msg, e, 0, num, len);
        if (r < 2) {
          A::add(t, x, el);
          fe_select<A>(x, odd == (r == 1), t, x);
        } else {
          A::add(t, y, el);
          fe_select<A>(y, !odd, t, y);
        }
      }
    });
    permutation_pair<F, A, WIN>(x, y, odd, pc, tab);
    fe_select<A>(dig, b + 1 == blocks, x, dig);
  }
  if (order) {
    if (!odd && s < cnt) store_digest<A>(out + item * (A::NABI / 4), dig);   // digest = state[0] = the even lane's x
    return;
  }
  __syncthreads();
  lds_put<A>(lds, odd ? kPairStates + s : s, dig);  // digest = state[0] = the even lane's x
  block_store<A::NABI / 4>(lds, out, st0, cnt);
}

// Merkle authentication: lane i hashes leaf i up its path (depth sibling digests, bottom-up) with
// the 2-1 instance's merge (= Jive compress of [left, right], anemoi_2_1/hasher.rs:87-92) and writes
// the recomputed root; bit l of index[i] says whether the node is the right child at level l.
template <int FIELD>
ANEMOI_KERNEL void k_merkle_climb(const uint4* __restrict__ leaves, const uint64_t* __restrict__ index,
                                  const uint4* __restrict__ paths, unsigned depth, size_t n, uint4* __restrict__ out,
                                  PermConsts pc) {
  using F = FieldC<FIELD>;
  using A = ArithFor<FIELD>;
  constexpr int WIN = KernelCfg<F::N>::WIN, Q = A::NABI / 4;
  extern __shared__ uint4 lds[];
  const size_t blk0 = size_t(blockIdx.x) * kBlock;
  const int cnt = n - blk0 < size_t(kBlock) ? int(n - blk0) : kBlock;
  const size_t item = blk0 + (threadIdx.x < cnt ? threadIdx.x : 0);  // idle lanes redo item blk0
  const uint64_t idx = index[item];
  auto load = [&](const uint4* src, typename A::Fe& v) {
    uint32_t w[A::NABI];
#pragma unroll
    for (int q = 0; q < Q; q++) {
      uint4 x = src[q];
      w[4 * q] = x.x, w[4 * q + 1] = x.y, w[4 * q + 2] = x.z, w[4 * q + 3] = x.w;
    }
    A::from_abi(v, w);
  };
  typename A::Fe cur, sib;
  load(leaves + item * Q, cur);
  const LdsTable<A> tab = make_table<A>(lds);
#pragma nounroll
  for (unsigned l = 0; l < depth; l++) {
    load(paths + (item * depth + l) * Q, sib);
    const bool right = (idx >> l) & 1;  // this node is the right child: state = [sibling, node]
    typename A::Fe st[2], sum;
#pragma unroll
    for (int i = 0; i < A::NL; i++) {
      st[0].l[i] = right ? sib.l[i] : cur.l[i];
      st[1].l[i] = right ? cur.l[i] : sib.l[i];
    }
    A::add(sum, cur, sib);
    // (round 2 ran the plain window here: with two 13-limb constants hoisted across the loop the extra digits
    // pushed this kernel to 182 VGPRs; since the constants are pinned next to their use it fits with the digits)
    permutation<F, A, 2, WIN>(st, pc, tab);
    A::add(cur, st[0], st[1]);
    A::add(cur, cur, sum);
    if (A::kLoose) A::settle(cur);
  }
  __syncthreads();
  lds_put<A>(lds, threadIdx.x, cur);
  block_store<Q>(lds, out, blk0, cnt);
}

}  // namespace anemoi
#include "anemoi_coop_kernels.h"   // the latency kernels (row-cooperative arithmetic) and their cut-offs
namespace anemoi {

// to = true: canonical -> Montgomery (x * R^2 / R); to = false: Montgomery -> canonical (x * 1 / R).
// Always on 32-bit limbs: this is the ABI's own R = 2^(32 N).
template <int FIELD>
__global__ __launch_bounds__(kBlock) void k_mont_convert(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                         size_t count, int to) {
  using F = FieldC<FIELD>;
  using A = Arith32<F>;
  constexpr int N = F::N;
  extern __shared__ uint4 lds[];
  const size_t blk0 = size_t(blockIdx.x) * kBlock;
  const int cnt = count - blk0 < size_t(kBlock) ? int(count - blk0) : kBlock;
  block_load<N / 4>(lds, in, blk0, cnt);
  Fe<N> v, k;
  lds_get<A>(lds, threadIdx.x, v);
#pragma unroll
  for (int i = 0; i < N; i++) k.l[i] = to ? F::R2[i] : (i == 0 ? 1u : 0u);
  mont_mul<F, true>(v, v, k);
  __syncthreads();
  lds_put<A>(lds, threadIdx.x, v);
  block_store<N / 4>(lds, out, blk0, cnt);
}

}  // namespace anemoi
#include "anemoi_generic.h"
namespace anemoi {

// ---- launchers (one set per field translation unit) ------------------------------------------------
#ifndef ANEMOI_BOUNDS_WALK

struct HostConsts {  // what the context uploads for one (field, width)
  std::vector<uint32_t> ark_c, ark_d;    // lane-private limb layout (F::Lane)
  std::vector<uint32_t> coop_c, coop_d;  // the instance's round constants in the cooperative kernels' layout (F::Coop)
  std::vector<uint32_t> fold_c, fold_d;  // ... and in the two-row fold layout (F::Fold)
  std::vector<uint8_t> sched, sched5, sched_plain;
  int steps, first, steps5, first5, steps_plain, first_plain;
};

struct FieldOps {
  int limbs64, chunk, rounds21, rounds43, generator, alpha;
  const char* name;
  void (*host_consts)(int width, HostConsts* out);
  hipError_t (*permutation)(int width, int sbox_only, void* d_states, size_t n, PermConsts pc, hipStream_t s);
  hipError_t (*jive)(int width, int k, const void* d_in, void* d_out, size_t n, PermConsts pc, hipStream_t s);
  hipError_t (*sponge)(int width, int bytes, const void* d_src, size_t per_msg, size_t n, void* d_out, PermConsts pc,
                       hipStream_t s);
  // one segment of every message (SpongeSeg): per_msg = this launch's bytes / elements per message
  hipError_t (*sponge_seg)(int width, int bytes, const void* d_src, size_t per_msg, size_t n, void* d_out, PermConsts pc,
                           SpongeSeg seg, hipStream_t s);
  // messages of different lengths: message i = bytes [off[i], off[i+1]) of d_msgs
  // (bytes: byte messages and byte offsets; else messages of ABI elements and offsets in elements)
  // d_order: null, or n 32-bit message indices (slot j works on message d_order[j] and writes out[d_order[j]])
  // d_status: null, or the bucketing's status word (non-zero: malformed offsets -- zero digests, no message byte read)
  hipError_t (*sponge_ragged)(int width, int bytes, const void* d_msgs, const void* d_off, size_t n, void* d_out, PermConsts pc,
                              const void* d_order, const void* d_status, hipStream_t s);
  hipError_t (*mont_convert)(int to, const void* d_in, void* d_out, size_t count, hipStream_t s);
  hipError_t (*merkle_climb)(const void* d_leaves, const void* d_index, const void* d_paths, unsigned depth, size_t n,
                             void* d_out, PermConsts pc, hipStream_t s);
  // run-time instances (anemoi_generic.h)
  hipError_t (*generic_permutation)(void* d_states, size_t n, GenericConsts gc, PermConsts pc, hipStream_t s);
  hipError_t (*generic_jive)(const void* d_in, void* d_out, size_t n, int k, GenericConsts gc, PermConsts pc,
                             hipStream_t s);
  hipError_t (*generic_sponge)(int bytes, const void* d_src, size_t per_msg, size_t n, void* d_out, int rate,
                               GenericConsts gc, PermConsts pc, hipStream_t s);
  hipError_t (*exp_alpha)(int inverse, void* d_elems, size_t n, PermConsts pc, hipStream_t s);
  // run-time-instance constants: ABI elements -> the kernels' internal form (generic_stride words each)
  hipError_t (*generic_prepare)(const void* d_abi, void* d_out, size_t count, hipStream_t s);
  int generic_stride;
  // items one full wave of workgroups of a batch kernel processes on the current device (every CU at its
  // resident-workgroup limit, from the occupancy API): the chunk quantum of the host-pointer pipeline
  size_t (*wave_items)(int kind, int width, int num_cus);
  // one small launch of every THROUGHPUT kernel of (field, width) -- Jive, permutation, sponge over bytes and over
  // elements -- on n items of zeros (d_buf: n states, d_out: n states of room): anemoi_warmup
  hipError_t (*warmup)(int width, void* d_buf, void* d_out, size_t n, PermConsts pc, hipStream_t s);
#if ANEMOI_AB_BUILD
  // Jive 2-1 with the blocks handed out by a counter (d_queue[0] = 0 at launch); `wgs` resident workgroups
  // wgs = 0: the shipped static dealing (k_jive_acct); d_acct: null or 8 XcdAcct records (see there)
  hipError_t (*jive_queue)(const void* d_in, void* d_out, size_t n, PermConsts pc, uint32_t* d_queue, unsigned wgs, void* d_acct, hipStream_t s);
#endif
};

enum KernelKind { kKindPermutation = 0, kKindJive = 1, kKindSponge = 2, kKindConvert = 3, kKindExpAlpha = 4 };

const FieldOps* field_ops(int field);  // capi.hip

// UNDERFILLED LAUNCHES GET A DO-NOTHING LAUNCH IN FRONT (round 6; tools/exp_cfg3_after_other_kernels.py, exp_balance_launch.py,
// exp_balance_launch2.py; profiles/r06/underfilled_launch_placement.txt).  A launch whose single-wavefront workgroups all fit
// the chip at once takes what its fullest SIMD takes, and the dispatcher does not place it evenly when it follows a launch that
// OVER-filled the chip (16 384 workgroups, of another kernel or of its own): eight workgroups on every CU as it should, but
// three wavefronts on ~30 of the 1 024 SIMDs and one on as many (tools/ubench/underfilled_after_overfull.hip records the HW_ID
// of every workgroup: profiles/r06/ubench_underfilled_after_overfull.txt).  Measured, every time: config 3 (2 048 workgroups resident for 333 ms) 485 ms, x 1.46; 1 024 workgroups
// x 1.84-1.89; 3 072 x 1.31; nothing at <= 512 workgroups or once the launch fills the chip itself.  Round 5 met the process's
// first launch of it and cured that one with anemoi_warmup; the rule is wider -- a headline launch and config 3 alternating:
// every config-3 launch slow.  What restores an even placement is ANY launch of <= 8 192 single-wavefront workgroups that
// all retire together, in between -- 512 of them asleep for 20 us do, 25 us in all (16 384 do not, nor do 256-thread
// workgroups; capping the workgroups per CU by their LDS request does not either: the imbalance is between the SIMDs of a
// CU).  So every launch of 2 ... 16 workgroups per CU is preceded by k_balance: four workgroups per CU, ~20 us.  Only a
// kernel launch on the caller's stream: capturable.  Option balance_underfilled = 0 switches it off (the A/B).
void balance_launch(unsigned wgs, hipStream_t s);   // capi.hip: k_balance (one copy for the seven field translation units)
inline void balance_before(size_t wgs, const PermConsts& pc, hipStream_t s) {
  const size_t cus = size_t(pc.simds) / 4;
  if (!cus || wgs <= 2 * cus || wgs >= 16 * cus || !opt::get_or(opt::kBalanceUnderfilled, 1)) return;
  balance_launch(unsigned(4 * cus), s);
}
inline unsigned grid_for(size_t n) { return unsigned((n + kBlock - 1) / kBlock); }
inline unsigned pair_grid(size_t n) { return unsigned((n + kBlock / 2 - 1) / (kBlock / 2)); }  // 32 states per workgroup

template <int FIELD>
struct Launch {
  using F = FieldC<FIELD>;
  using A = ArithFor<FIELD>;
  static constexpr int N = F::N, WIN = KernelCfg<N>::WIN;

  static void host_consts(int width, HostConsts* hc) {
    const uint32_t* c = A::host_ark(width, false);
    const uint32_t* d = A::host_ark(width, true);
    const int cnt = (width == 2 ? F::kRounds21 : 2 * F::kRounds43) * A::NL;
    hc->ark_c.assign(c, c + cnt);
    hc->ark_d.assign(d, d + cnt);
    using CL = typename F::Coop;
    using FL = typename F::Fold;
    if (width == 2) {
      hc->coop_c.assign(CL::ArkC_21, CL::ArkC_21 + F::kRounds21 * CL::NL);
      hc->coop_d.assign(CL::ArkD_21, CL::ArkD_21 + F::kRounds21 * CL::NL);
      hc->fold_c.assign(FL::ArkC_21, FL::ArkC_21 + F::kRounds21 * FL::NL);
      hc->fold_d.assign(FL::ArkD_21, FL::ArkD_21 + F::kRounds21 * FL::NL);
    } else {
      hc->coop_c.assign(CL::ArkC_43, CL::ArkC_43 + 2 * F::kRounds43 * CL::NL);
      hc->coop_d.assign(CL::ArkD_43, CL::ArkD_43 + 2 * F::kRounds43 * CL::NL);
      hc->fold_c.assign(FL::ArkC_43, FL::ArkC_43 + 2 * F::kRounds43 * FL::NL);
      hc->fold_d.assign(FL::ArkD_43, FL::ArkD_43 + 2 * F::kRounds43 * FL::NL);
    }
    static_assert(WIN >= 2 && WIN <= 5, "schedules are generated for 2..5-bit windows");
    static_assert(F::kCoopWin >= 2 && F::kCoopWin <= 5, "");
    if (F::kCoopWin == 5) {
      hc->sched5.assign(F::kW5Sched, F::kW5Sched + 2 * F::kW5Steps);
      hc->steps5 = F::kW5Steps, hc->first5 = F::kW5First;
    } else if (F::kCoopWin == 4) {
      hc->sched5.assign(F::kW4Sched, F::kW4Sched + 2 * F::kW4Steps);
      hc->steps5 = F::kW4Steps, hc->first5 = F::kW4First;
    } else if (F::kCoopWin == 3) {
      hc->sched5.assign(F::kW3Sched, F::kW3Sched + 2 * F::kW3Steps);
      hc->steps5 = F::kW3Steps, hc->first5 = F::kW3First;
    } else {
      hc->sched5.assign(F::kW2Sched, F::kW2Sched + 2 * F::kW2Steps);
      hc->steps5 = F::kW2Steps, hc->first5 = F::kW2First;
    }
    if (WIN == 2) {
      hc->sched.assign(F::kW2Sched, F::kW2Sched + 2 * F::kW2Steps);
      hc->steps = F::kW2Steps, hc->first = F::kW2First;
    } else if (WIN == 3) {
      hc->sched.assign(F::kW3Sched, F::kW3Sched + 2 * F::kW3Steps);
      hc->steps = F::kW3Steps, hc->first = F::kW3First;
    } else if (WIN == 4) {
      hc->sched.assign(F::kW4Sched, F::kW4Sched + 2 * F::kW4Steps);
      hc->steps = F::kW4Steps, hc->first = F::kW4First;
    } else {
      hc->sched.assign(F::kW5Sched, F::kW5Sched + 2 * F::kW5Steps);
      hc->steps = F::kW5Steps, hc->first = F::kW5First;
    }
    hc->sched_plain = hc->sched, hc->steps_plain = hc->steps, hc->first_plain = hc->first;
    if (WIN == 3 && F::kXDigits > 0 && ANEMOI_XDIGITS_ON) {  // window 3 + extra digits in VGPRs (anemoi_perm.h)
      hc->sched.assign(F::kXSched, F::kXSched + 2 * F::kXSteps);
      hc->steps = F::kXSteps, hc->first = F::kXFirst;
    }
  }

  static hipError_t permutation(int width, int sbox_only, void* d, size_t n, PermConsts pc, hipStream_t s) {
    if (!n) return hipSuccess;
    // latency path: the cut-offs of the Jive kernels (same permutation, same items per wavefront)
    if (!sbox_only && width == 2 && n <= coop2d_max_items(pc.simds)) {
      const size_t groups = (n + 1) / 2;
      const unsigned g = groups < 65536 ? unsigned(groups) : 65536u;
      balance_before(g, pc, s);
      k_permutation_coop<FIELD, 2, 32><<<g, kBlock, 0, s>>>((uint32_t*)d, n, pc);
      return hipGetLastError();
    }
    if (!sbox_only && width == 4 && n <= coop2d43_max_items(pc.simds)) {   // one 4-3 state per wavefront, a column per row pair
      const unsigned g = n < 65536 ? unsigned(n) : 65536u;
      balance_before(g, pc, s);
      k_permutation_coop<FIELD, 4, 32><<<g, kBlock, 0, s>>>((uint32_t*)d, n, pc);
      return hipGetLastError();
    }
    if (!sbox_only && n <= (width == 2 ? coop4_max_items(pc.simds) : coop43_max_items(pc.simds, F::Coop::NL))) {
      const size_t groups = width == 2 ? (n + 3) / 4 : (n + 1) / 2;
      const unsigned g = groups < 65536 ? unsigned(groups) : 65536u;
      balance_before(g, pc, s);
      if (width == 2) k_permutation_coop<FIELD, 2><<<g, kBlock, 0, s>>>((uint32_t*)d, n, pc);
      else k_permutation_coop<FIELD, 4><<<g, kBlock, 0, s>>>((uint32_t*)d, n, pc);
      return hipGetLastError();
    }
    balance_before(width == 2 ? grid_for(n) : pair_grid(n), pc, s);
    if (width == 2 && !sbox_only)
      k_permutation<FIELD, 2, false><<<grid_for(n), kBlock, lds_bytes<A, WIN, 2>(), s>>>((uint4*)d, n, pc);
    else if (width == 2)
      k_permutation<FIELD, 2, true><<<grid_for(n), kBlock, lds_bytes<A, WIN, 2>(), s>>>((uint4*)d, n, pc);
    else if (!sbox_only)
      k_permutation_pair<FIELD, false><<<pair_grid(n), kBlock, lds_bytes<A, WIN, 2>(), s>>>((uint4*)d, n, pc);
    else
      k_permutation_pair<FIELD, true><<<pair_grid(n), kBlock, lds_bytes<A, WIN, 2>(), s>>>((uint4*)d, n, pc);
    return hipGetLastError();
  }

  static hipError_t jive(int width, int k, const void* in, void* out, size_t n, PermConsts pc, hipStream_t s) {
    if (!n) return hipSuccess;
#if ANEMOI_AB_BUILD   // the recorded negative: one item per wavefront (four-row fold on 11 limbs, else rounds 1-2's scan)
    if (width == 2 && n <= coop_max_items()) {
      const unsigned g = n < 65536 ? unsigned(n) : 65536u;  // the kernel strides over items
      balance_before(g, pc, s);
      k_jive2_coop<FIELD, 64><<<g, kBlock, 0, s>>>((const uint32_t*)in, (uint32_t*)out, n, pc);
      return hipGetLastError();
    }
#endif
    if (width == 2 && n <= coop2d_max_items(pc.simds)) {  // two items per wavefront on row pairs: the lowest latency
      const size_t groups = (n + 1) / 2;
      const unsigned g = groups < 65536 ? unsigned(groups) : 65536u;
      balance_before(g, pc, s);
      k_jive2_coop<FIELD, 32><<<g, kBlock, 0, s>>>((const uint32_t*)in, (uint32_t*)out, n, pc);
      return hipGetLastError();
    }
    if (width == 2 && n <= coop4_max_items(pc.simds)) {  // four items per wavefront, one per DPP row
      const size_t groups = (n + 3) / 4;
      const unsigned g = groups < 65536 ? unsigned(groups) : 65536u;
      balance_before(g, pc, s);
      k_jive2_coop<FIELD, 16><<<g, kBlock, 0, s>>>((const uint32_t*)in, (uint32_t*)out, n, pc);
      return hipGetLastError();
    }
    if (width == 4 && n <= coop2d43_max_items(pc.simds)) {  // 4-3, lowest latency: one state per wavefront on the two-row fold
      const unsigned g = n < 65536 ? unsigned(n) : 65536u;
      balance_before(g, pc, s);
      if (k == 2) k_jive4_coop<FIELD, 2, 32><<<g, kBlock, 0, s>>>((const uint32_t*)in, (uint32_t*)out, n, pc);
      else k_jive4_coop<FIELD, 4, 32><<<g, kBlock, 0, s>>>((const uint32_t*)in, (uint32_t*)out, n, pc);
      return hipGetLastError();
    }
    if (width == 4 && n <= coop43_max_items(pc.simds, F::Coop::NL)) {  // 4-3 latency path: two states per wavefront
      const size_t groups = (n + 1) / 2;
      const unsigned g = groups < 65536 ? unsigned(groups) : 65536u;
      balance_before(g, pc, s);
      if (k == 2) k_jive4_coop<FIELD, 2><<<g, kBlock, 0, s>>>((const uint32_t*)in, (uint32_t*)out, n, pc);
      else k_jive4_coop<FIELD, 4><<<g, kBlock, 0, s>>>((const uint32_t*)in, (uint32_t*)out, n, pc);
      return hipGetLastError();
    }
    balance_before(width == 2 ? grid_for(n) : pair_grid(n), pc, s);
    if (width == 2)
      k_jive<FIELD, 2, 2><<<grid_for(n), kBlock, lds_bytes<A, WIN, 2>(), s>>>((const uint4*)in, (uint4*)out, n, pc);
    else if (k == 2)
      k_jive_pair<FIELD, 2><<<pair_grid(n), kBlock, lds_bytes<A, WIN, 2>(), s>>>((const uint4*)in, (uint4*)out, n, pc);
    else
      k_jive_pair<FIELD, 4><<<pair_grid(n), kBlock, lds_bytes<A, WIN, 2>(), s>>>((const uint4*)in, (uint4*)out, n, pc);
    return hipGetLastError();
  }

  static hipError_t sponge_seg(int width, int bytes, const void* src, size_t per_msg, size_t n, void* out, PermConsts pc,
                               SpongeSeg seg, hipStream_t s) {
    if (!n) return hipSuccess;
    // small batches: the latency kernels, whole messages and segments alike (the state is carried in seg.state)
    if (width == 2 && n <= coop2d_max_items(pc.simds)) {  // two messages per wavefront
      const size_t groups = (n + 1) / 2;
      const unsigned g = groups < 65536 ? unsigned(groups) : 65536u;
      balance_before(g, pc, s);
      if (bytes) k_sponge_coop<FIELD, 2, true, 32><<<g, kBlock, 0, s>>>(src, per_msg, n, (uint32_t*)out, pc, seg);
      else k_sponge_coop<FIELD, 2, false, 32><<<g, kBlock, 0, s>>>(src, per_msg, n, (uint32_t*)out, pc, seg);
      return hipGetLastError();
    }
    if (width == 4 && n <= coop2d43_max_items(pc.simds)) {  // one 4-3 message per wavefront
      const unsigned g = n < 65536 ? unsigned(n) : 65536u;
      balance_before(g, pc, s);
      if (bytes) k_sponge_coop<FIELD, 4, true, 32><<<g, kBlock, 0, s>>>(src, per_msg, n, (uint32_t*)out, pc, seg);
      else k_sponge_coop<FIELD, 4, false, 32><<<g, kBlock, 0, s>>>(src, per_msg, n, (uint32_t*)out, pc, seg);
      return hipGetLastError();
    }
    if (n <= coop_sponge_max_items(pc.simds)) {  // four (2-1) / two (4-3) messages per wavefront on the scan
      const size_t groups = width == 2 ? (n + 3) / 4 : (n + 1) / 2;
      const unsigned g = groups < 65536 ? unsigned(groups) : 65536u;
      balance_before(g, pc, s);
      if (width == 2 && bytes) k_sponge_coop<FIELD, 2, true><<<g, kBlock, 0, s>>>(src, per_msg, n, (uint32_t*)out, pc, seg);
      else if (width == 2) k_sponge_coop<FIELD, 2, false><<<g, kBlock, 0, s>>>(src, per_msg, n, (uint32_t*)out, pc, seg);
      else if (bytes) k_sponge_coop<FIELD, 4, true><<<g, kBlock, 0, s>>>(src, per_msg, n, (uint32_t*)out, pc, seg);
      else k_sponge_coop<FIELD, 4, false><<<g, kBlock, 0, s>>>(src, per_msg, n, (uint32_t*)out, pc, seg);
      return hipGetLastError();
    }
    const size_t l = lds_bytes<A, WIN, 1>(), lp = lds_bytes<A, WIN, 2>();
    balance_before(width == 2 ? grid_for(n) : pair_grid(n), pc, s);
    if (width == 2 && bytes)
      k_sponge<FIELD, 2, true><<<grid_for(n), kBlock, l, s>>>(src, per_msg, n, (uint4*)out, pc, seg);
    else if (width == 2)
      k_sponge<FIELD, 2, false><<<grid_for(n), kBlock, l, s>>>(src, per_msg, n, (uint4*)out, pc, seg);
    else if (bytes)
      k_sponge_pair<FIELD, true><<<pair_grid(n), kBlock, lp, s>>>(src, per_msg, n, (uint4*)out, pc, seg);
    else
      k_sponge_pair<FIELD, false><<<pair_grid(n), kBlock, lp, s>>>(src, per_msg, n, (uint4*)out, pc, seg);
    return hipGetLastError();
  }

  static hipError_t sponge(int width, int bytes, const void* src, size_t per_msg, size_t n, void* out, PermConsts pc,
                           hipStream_t s) {
    return sponge_seg(width, bytes, src, per_msg, n, out, pc, SpongeSeg{nullptr, 0, per_msg, 1, 1}, s);
  }

  static hipError_t sponge_ragged(int width, int bytes, const void* msgs, const void* off, size_t n, void* out, PermConsts pc,
                                  const void* order, const void* status, hipStream_t s) {
    if (!n) return hipSuccess;
    return bytes ? sponge_ragged_as<true>(width, msgs, off, n, out, pc, order, status, s)
                 : sponge_ragged_as<false>(width, msgs, off, n, out, pc, order, status, s);
  }
  template <bool BYTES>
  static hipError_t sponge_ragged_as(int width, const void* msgs, const void* off, size_t n, void* out, PermConsts pc,
                                     const void* order, const void* status, hipStream_t s) {
    // small batches: the latency kernels, at the cut-offs of the equal-length sponge (sponge_seg above)
    const uint8_t* m = (const uint8_t*)msgs;
    const uint64_t* o = (const uint64_t*)off;
    const uint32_t* ord = (const uint32_t*)order;
    const uint32_t* st = (const uint32_t*)status;
    if (width == 2 && n <= coop2d_max_items(pc.simds)) {  // two messages per wavefront
      const size_t groups = (n + 1) / 2;
      const unsigned g = groups < 65536 ? unsigned(groups) : 65536u;
      balance_before(g, pc, s);
      k_sponge_ragged_coop<FIELD, 2, BYTES, 32><<<g, kBlock, 0, s>>>(m, o, n, (uint32_t*)out, pc, ord, st);
      return hipGetLastError();
    }
    if (width == 4 && n <= coop2d43_max_items(pc.simds)) {  // one 4-3 message per wavefront
      const unsigned g = n < 65536 ? unsigned(n) : 65536u;
      balance_before(g, pc, s);
      k_sponge_ragged_coop<FIELD, 4, BYTES, 32><<<g, kBlock, 0, s>>>(m, o, n, (uint32_t*)out, pc, ord, st);
      return hipGetLastError();
    }
    if (n <= coop_sponge_max_items(pc.simds)) {  // four (2-1) / two (4-3) messages per wavefront on the scan
      const size_t groups = width == 2 ? (n + 3) / 4 : (n + 1) / 2;
      const unsigned g = unsigned(groups < 65536 ? groups : 65536);
      balance_before(g, pc, s);
      if (width == 2) k_sponge_ragged_coop<FIELD, 2, BYTES, 16><<<g, kBlock, 0, s>>>(m, o, n, (uint32_t*)out, pc, ord, st);
      else k_sponge_ragged_coop<FIELD, 4, BYTES, 16><<<g, kBlock, 0, s>>>(m, o, n, (uint32_t*)out, pc, ord, st);
      return hipGetLastError();
    }
    balance_before(width == 2 ? grid_for(n) : pair_grid(n), pc, s);
    if (width == 2)
      k_sponge_ragged<FIELD, BYTES><<<grid_for(n), kBlock, lds_bytes<A, WIN, 1>(), s>>>(m, o, n, (uint4*)out, pc, ord, st);
    else
      k_sponge_ragged_pair<FIELD, BYTES><<<pair_grid(n), kBlock, lds_bytes<A, WIN, 2>(), s>>>(m, o, n, (uint4*)out, pc, ord, st);
    return hipGetLastError();
  }

  static hipError_t mont_convert(int to, const void* in, void* out, size_t count, hipStream_t s) {
    if (!count) return hipSuccess;
    k_mont_convert<FIELD><<<grid_for(count), kBlock, size_t(N) * 4 * kBlock, s>>>((const uint4*)in, (uint4*)out, count, to);
    return hipGetLastError();
  }

  static hipError_t merkle_climb(const void* leaves, const void* index, const void* paths, unsigned depth, size_t n,
                                 void* out, PermConsts pc, hipStream_t s) {
    if (!n) return hipSuccess;
    if (n <= coop2d_max_items(pc.simds)) {  // very few paths: two per wavefront on row pairs
      const size_t groups = (n + 1) / 2;
      const unsigned g = groups < 65536 ? unsigned(groups) : 65536u;
      balance_before(g, pc, s);
      k_merkle_climb_coop<FIELD, 32><<<g, kBlock, 0, s>>>((const uint32_t*)leaves, (const uint64_t*)index,
                                                           (const uint32_t*)paths, depth, n, (uint32_t*)out, pc);
      return hipGetLastError();
    }
    if (n <= coop_climb_max_items(pc.simds)) {  // a handful of paths: the latency form
      const size_t groups = (n + 3) / 4;
      const unsigned g = groups < 65536 ? unsigned(groups) : 65536u;
      balance_before(g, pc, s);
      k_merkle_climb_coop<FIELD><<<g, kBlock, 0, s>>>((const uint32_t*)leaves, (const uint64_t*)index,
                                                       (const uint32_t*)paths, depth, n, (uint32_t*)out, pc);
      return hipGetLastError();
    }
    balance_before(grid_for(n), pc, s);
    k_merkle_climb<FIELD><<<grid_for(n), kBlock, lds_bytes<A, WIN, 1>(), s>>>(
        (const uint4*)leaves, (const uint64_t*)index, (const uint4*)paths, depth, n, (uint4*)out, pc);
    return hipGetLastError();
  }

  static hipError_t generic_permutation(void* d, size_t n, GenericConsts gc, PermConsts pc, hipStream_t s) {
    if (!n) return hipSuccess;
    balance_before(cols_grid(n, gc.cols), pc, s);
    k_permutation_cols<FIELD><<<cols_grid(n, gc.cols), kBlock, lds_bytes<A, WIN, 1>(), s>>>((uint32_t*)d, n, gc, pc);
    return hipGetLastError();
  }

  static hipError_t generic_jive(const void* in, void* out, size_t n, int k, GenericConsts gc, PermConsts pc,
                                 hipStream_t s) {
    if (!n) return hipSuccess;
    balance_before(cols_grid(n, gc.cols), pc, s);
    k_jive_cols<FIELD><<<cols_grid(n, gc.cols), kBlock, lds_bytes<A, WIN, 1>(), s>>>((const uint32_t*)in, (uint32_t*)out,
                                                                                      n, k, gc, pc);
    return hipGetLastError();
  }

  static hipError_t generic_sponge(int bytes, const void* src, size_t per_msg, size_t n, void* out, int rate,
                                   GenericConsts gc, PermConsts pc, hipStream_t s) {
    if (!n) return hipSuccess;
    const unsigned g = cols_grid(n, gc.cols);
    balance_before(g, pc, s);
    if (bytes)
      k_sponge_cols<FIELD, true><<<g, kBlock, lds_bytes<A, WIN, 1>(), s>>>(src, per_msg, n, (uint32_t*)out, rate, gc, pc);
    else
      k_sponge_cols<FIELD, false><<<g, kBlock, lds_bytes<A, WIN, 1>(), s>>>(src, per_msg, n, (uint32_t*)out, rate, gc, pc);
    return hipGetLastError();
  }

  static hipError_t exp_alpha(int inverse, void* d, size_t n, PermConsts pc, hipStream_t s) {
    if (!n) return hipSuccess;
    balance_before(grid_for(n), pc, s);
    if (inverse) k_exp_alpha<FIELD, true><<<grid_for(n), kBlock, lds_bytes<A, WIN, 1>(), s>>>((uint4*)d, n, pc);
    else k_exp_alpha<FIELD, false><<<grid_for(n), kBlock, lds_bytes<A, WIN, 1>(), s>>>((uint4*)d, n, pc);
    return hipGetLastError();
  }

  static hipError_t generic_prepare(const void* abi, void* out, size_t count, hipStream_t s) {
    if (!count) return hipSuccess;
    k_generic_prepare<FIELD><<<grid_for(count), kBlock, 0, s>>>((const uint32_t*)abi, (uint32_t*)out, count);
    return hipGetLastError();
  }

  // The FIRST big dispatch of a process can spread its wavefronts unevenly over the SIMDs of a CU (3 on one, 1 on
  // another where 2 + 2 fit: tools/ubench/first_launch_placement.hip, profiles/r05/ubench_first_launch_placement.txt); a
  // kernel whose workgroups all start at once and run for the whole launch -- config 3: 2 048 wavefronts, 330 ms -- then
  // takes x 1.5.  A launch of the same kernel that reaches every CU beforehand cures it (profiles/r04/
  // first_launch_after_idle.txt: 4 096 messages do, 64 do not).  So: every lane-private kernel of the instance once, on
  // n >= 16 x SIMDs items (>= 256 workgroups), one-element messages for the sponges.
  static hipError_t warmup(int width, void* d_buf, void* d_out, size_t n, PermConsts pc, hipStream_t s) {
    const SpongeSeg whole_b{nullptr, 0, size_t(F::kChunk), 1, 1}, whole_e{nullptr, 0, 1, 1, 1};
    const size_t l1 = lds_bytes<A, WIN, 1>(), l2 = lds_bytes<A, WIN, 2>();
    if (width == 2) {
      k_jive<FIELD, 2, 2><<<grid_for(n), kBlock, l2, s>>>((const uint4*)d_buf, (uint4*)d_out, n, pc);
      k_permutation<FIELD, 2, false><<<grid_for(n), kBlock, l2, s>>>((uint4*)d_out, n, pc);
      k_sponge<FIELD, 2, true><<<grid_for(n), kBlock, l1, s>>>(d_buf, size_t(F::kChunk), n, (uint4*)d_out, pc, whole_b);
      k_sponge<FIELD, 2, false><<<grid_for(n), kBlock, l1, s>>>(d_buf, 1, n, (uint4*)d_out, pc, whole_e);
    } else {
      k_jive_pair<FIELD, 2><<<pair_grid(n), kBlock, l2, s>>>((const uint4*)d_buf, (uint4*)d_out, n, pc);
      k_permutation_pair<FIELD, false><<<pair_grid(n), kBlock, l2, s>>>((uint4*)d_out, n, pc);
      k_sponge_pair<FIELD, true><<<pair_grid(n), kBlock, l2, s>>>(d_buf, size_t(F::kChunk), n, (uint4*)d_out, pc, whole_b);
      k_sponge_pair<FIELD, false><<<pair_grid(n), kBlock, l2, s>>>(d_buf, 1, n, (uint4*)d_out, pc, whole_e);
    }
    return hipGetLastError();
  }

#if ANEMOI_AB_BUILD
  static hipError_t jive_queue(const void* in, void* out, size_t n, PermConsts pc, uint32_t* queue, unsigned wgs, void* acct, hipStream_t s) {
    XcdAcct* a = (XcdAcct*)acct;
    if (wgs == 0)   // the shipped dealing, accounted
      k_jive_acct<FIELD, 2, 2><<<grid_for(n), kBlock, lds_bytes<A, WIN, 2>(), s>>>((const uint4*)in, (uint4*)out, n, pc, a);
    else if (wgs >= grid_for(n))   // at least one workgroup per block: the ticket form (one block per workgroup, surplus workgroups leave)
      k_jive_ticket<FIELD, 2, 2><<<wgs, kBlock, lds_bytes<A, WIN, 2>(), s>>>((const uint4*)in, (uint4*)out, n, pc, queue, a);
    else
      k_jive_queue<FIELD, 2, 2><<<wgs, kBlock, lds_bytes<A, WIN, 2>(), s>>>((const uint4*)in, (uint4*)out, n, pc, queue, a);
    return hipGetLastError();
  }
#endif

  template <class K>
  static size_t resident_items(K kernel, size_t lds, int items_per_wg, int num_cus) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kBlock, lds) != hipSuccess || nb <= 0) {
      (void)hipGetLastError();
      nb = 12;  // 3 waves per SIMD: what the 13/14-limb kernels are built for
    }
    return size_t(nb) * size_t(num_cus) * size_t(items_per_wg);
  }

  static size_t wave_items(int kind, int width, int num_cus) {
    switch (kind) {
      case kKindPermutation:
        return width == 2 ? resident_items(k_permutation<FIELD, 2, false>, lds_bytes<A, WIN, 2>(), kBlock, num_cus)
                          : resident_items(k_permutation_pair<FIELD, false>, lds_bytes<A, WIN, 2>(), kPairStates, num_cus);
      case kKindJive:
        return width == 2 ? resident_items(k_jive<FIELD, 2, 2>, lds_bytes<A, WIN, 2>(), kBlock, num_cus)
                          : resident_items(k_jive_pair<FIELD, 2>, lds_bytes<A, WIN, 2>(), kPairStates, num_cus);
      case kKindSponge:
        return width == 2 ? resident_items(k_sponge<FIELD, 2, true>, lds_bytes<A, WIN, 1>(), kBlock, num_cus)
                          : resident_items(k_sponge_pair<FIELD, true>, lds_bytes<A, WIN, 2>(), kPairStates, num_cus);
      case kKindExpAlpha: return resident_items(k_exp_alpha<FIELD, true>, lds_bytes<A, WIN, 1>(), kBlock, num_cus);
      default: return resident_items(k_mont_convert<FIELD>, size_t(N) * 4 * kBlock, kBlock, num_cus);
    }
  }

  static const FieldOps* ops() {
    static const FieldOps o{F::L64,       F::kChunk,    F::kRounds21,        F::kRounds43, F::kG, F::kAlpha, F::kName,
                            host_consts,  permutation,  jive,                sponge,       sponge_seg,   sponge_ragged, mont_convert,
                            merkle_climb, generic_permutation, generic_jive, generic_sponge, exp_alpha,
                            generic_prepare, generic_stride<A>(), wave_items, warmup
#if ANEMOI_AB_BUILD
                            , jive_queue
#endif
    };
    return &o;
  }
};

#endif  // ANEMOI_BOUNDS_WALK
}  // namespace anemoi
