// coop29.h -- wave-cooperative Montgomery arithmetic: one 29-bit limb per lane, ONE field element per
// wavefront (LPR = 64) or FOUR, one per 16-lane DPP row (LPR = 16: lane 16 r + j holds limb j of element r).  This is the low-latency path (small batches, the top levels of a Merkle tree, a
// single `Jive::compress` through the shim); the lane-private path of mont29.h is the throughput
// path (4x the multiplications per instruction, but one lane needs ~2.6 ms / 8.6 ms for a
// 255-bit / 381-bit compression).
//
// Layout: lane j < NL holds limb j (same radix-2^29, R' = 2^(29 NL) Montgomery form as mont29.h);
// lanes >= NL hold zero.  A Montgomery product is a systolic operand scan over the lanes:
//     for i in 0..NL-1:
//         t_j += a_i * b_j                     a_i broadcast with v_readlane (SGPR operand)
//         m    = -(t_0) / p  mod 2^29          lane 0's low word -> SGPR, the multiply runs on the SALU
//         t_j += m * p_j                       now t_0 = 0 mod 2^29
//         t_j  = t_{j+1} (+ t_0 >> 29 in lane 0)   one DPP row_shl per accumulator word
// followed by two DPP carry passes that leave limbs < 2^29 + 2^7 (good enough as multiplier input:
// the column bound of mont29.h has a full bit of slack).  Where exact limbs are needed (before a
// subtraction, before the final comparison with p) the remaining single-bit carries are resolved
// with two wavefront ballots and one 64-bit scalar add -- carry look-ahead on the lane masks:
//     G = ballot(limb >= 2^29), P = ballot(limb == 2^29 - 1), carry-in mask C = ((G << 1) + P) ^ P
// and `v_addc_co_u32` consumes C directly as its per-lane carry-in.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include <type_traits>

#include "field_consts_gen.h"
#include "mont29.h"  // ANEMOI_ASM_MUL and the generated assembly (mont29_asm_gen.h)

namespace anemoi {

template <class F, int LPR = 64>
struct Coop29 {
  static_assert(LPR == 64 || LPR == 16, "one element per wavefront, or one per 16-lane DPP row");
  using L = typename F::Coop;  // always the 29-bit layout: the systolic scan below sums a whole
                               // column (2 NL products) in one 64-bit lane accumulator
  static_assert(L::W == 29, "column sums of 2 NL limb products must fit 64 bits");
  static constexpr int NL = L::NL, W = 29;
  static constexpr int NABI = F::N;
  static constexpr int kLanesPerItem = LPR;
  static constexpr uint32_t MASK = (1u << 29) - 1;
  static constexpr bool kTight = L::kTight;

  // DPP within row 0 (NL <= 14 < 16 lanes): out-of-range sources read as 0 (bound_ctrl)
  __device__ static __forceinline__ uint32_t from_next(uint32_t v) {  // lane j <- lane j+1
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101 /* row_shl:1 */, 0xf, 0xf, true);
  }
  __device__ static __forceinline__ uint32_t from_prev(uint32_t v) {  // lane j <- lane j-1
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
  }
  __device__ static __forceinline__ uint32_t lane() { return threadIdx.x; }
  __device__ static __forceinline__ uint32_t limb() { return threadIdx.x & (LPR - 1); }       // j: which limb this lane holds
  __device__ static __forceinline__ uint32_t row0() { return threadIdx.x & ~uint32_t(LPR - 1); }  // first lane of the element
  __device__ static __forceinline__ uint32_t keep(uint32_t v) { return limb() < NL ? v : 0u; }
  __device__ static __forceinline__ bool writer() { return true; }   // every lane of an element stores its own word

  // per-lane copy of a field constant (limb j, 0 beyond NL)
  __device__ static __forceinline__ uint32_t konst(const uint32_t* __restrict__ k) {
    return limb() < NL ? k[limb()] : 0u;
  }
  // The field constants a kernel needs, fetched ONCE (one vector load each): p, the subtraction pad, delta, R' mod p
  // (settle), g R' (mul_g of the tight fields) and the two ABI conversion factors.  Fetched where they are used -- as
  // rounds 1-2 did for One / GMont / In / Out -- a lone wavefront sits out a vector-memory round trip in front of
  // every settle() and mul_g(): four per round.
  struct K {
    uint32_t pl, kpl, delta, one, gm, in, out, rr;
  };
  __device__ static __forceinline__ K load_consts() {
    return K{konst(L::P), konst(L::KP), konst(L::Delta), konst(L::One), konst(L::GMont), konst(L::In), konst(L::Out),
             konst(L::RR)};
  }

  // two carry passes: 64-bit column sums -> limbs < 2^29 + 2^7
  __device__ static __forceinline__ uint32_t settle_columns(uint64_t t) {
    const uint32_t lo = (uint32_t)t & MASK;
    const uint64_t hi = t >> 29;  // < 2^35
    const uint64_t up = ((uint64_t)from_prev((uint32_t)(hi >> 32)) << 32) | from_prev((uint32_t)hi);
    const uint64_t r = (uint64_t)lo + up;  // < 2^36
    const uint32_t lo2 = (uint32_t)r & MASK;
    const uint32_t c = (uint32_t)(r >> 29);  // < 2^7
    return keep(lo2 + from_prev(c));
  }

  // one carry pass for sums of a few almost-normalised limbs (values < 2^32)
  __device__ static __forceinline__ uint32_t carry32(uint32_t r) {
    return keep((r & MASK) + from_prev(r >> 29));
  }

  // limbs < 2^30 -> limbs < 2^29 exactly (ballot carry look-ahead, see the header comment).  With four elements
  // per wavefront the one 64-bit add serves all four: lanes j >= NL of a row (at least two: NL <= 14) hold zero,
  // so neither generate nor propagate, and no carry crosses into the next row.
  __device__ static __forceinline__ uint32_t norm_exact(uint32_t r) {
    r = carry32(r);  // now < 2^29 + 8: at most a single carry bit out of any limb
    const unsigned long long G = __ballot(r > MASK), P = __ballot(r == MASK);
    const unsigned long long C = ((G << 1) + P) ^ P;
    return (r + (uint32_t)((C >> lane()) & 1)) & MASK;
  }

  // Montgomery product (limbs of a, b < 2^29 + 2^7): result < 2p when (a/p)(b/p) <= R'/p
  __device__ static __forceinline__ uint32_t mul(uint32_t a, uint32_t b, uint32_t pl) {
    const uint32_t sh = limb() == 0 ? 29u : 63u;
#if ANEMOI_ASM_MUL
    // the same scan as below, hand-scheduled (tools/gen_asm_mul.py gen_coop_mul / gen_coop4_mul: 10 issue slots
    // per step instead of hipcc's 13)
    if constexpr (LPR == 64) return settle_columns(AsmCoop<F::kId>::mul(a, b, pl, sh));
    else return settle_columns(AsmCoop4<F::kId>::mul(a, b, pl, sh, MASK));
#endif
    uint64_t t = 0;
    static_for_limbs<0>([&](auto I) {
      constexpr int i = decltype(I)::value;
      // a_i and the quotient digit: wave-uniform (v_readlane -> SGPR) with one element per wavefront, a DPP
      // broadcast inside each row with four
      uint32_t ai, t0;
      if constexpr (LPR == 64) ai = __builtin_amdgcn_readlane(a, i);
      else ai = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x150 + i /* row_newbcast:i */, 0xf, 0xf, false);
      t += (uint64_t)ai * b;
      if constexpr (LPR == 64) t0 = __builtin_amdgcn_readlane((uint32_t)t, 0);
      else t0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)t, 0x150 /* row_newbcast:0 */, 0xf, 0xf, false);
      const uint32_t m = (t0 * L::kN0Inv) & MASK;
      t += (uint64_t)m * pl;
      // lane 0's column is now 0 mod 2^29 and retires: its upper bits are the carry into the next
      // column.  Every column sum stays < 2^63, so shifting by 63 yields 0 in all other lanes:
      // a per-lane shift amount replaces a mask.
      const uint64_t u = t >> sh;
      t = (((uint64_t)from_next((uint32_t)(t >> 32)) << 32) | from_next((uint32_t)t)) + u;
    });
    return settle_columns(t);
  }

  template <int I, class Fn>
  __device__ static __forceinline__ void static_for_limbs(Fn&& fn) {
    if constexpr (I < NL) {
      fn(std::integral_constant<int, I>{});
      static_for_limbs<I + 1>(fn);
    }
  }

  // the forms the kernels call (the same on Coop2d, whose product needs more than the modulus)
  __device__ static __forceinline__ uint32_t mul(uint32_t a, uint32_t b, const K& k) { return mul(a, b, k.pl); }
  __device__ static __forceinline__ uint32_t sub(uint32_t a, uint32_t b, const K& k) { return sub(a, b, k.kpl); }
  __device__ static __forceinline__ uint32_t to_mont(uint32_t x, const K& k) { return mul(x, k.rr, k.pl); }
  __device__ static __forceinline__ uint32_t sqr_mul(uint32_t a, uint32_t n, uint32_t b, const K& k) {   // a^(2^n) b
    return mul(sqr_n(a, n, k), b, k.pl);
  }
  __device__ static __forceinline__ uint32_t sqr_n(uint32_t a, uint32_t n, const K& k) {   // a^(2^n)
#pragma nounroll
    for (uint32_t i = 0; i < n; i++) a = mul(a, a, k.pl);
    return a;
  }

  __device__ static __forceinline__ uint32_t add(uint32_t a, uint32_t b) { return carry32(a + b); }

  // a - b + kSubK29 * p (b < ~60 p, or < 2p for the tight fields); KP29 limbs are padded to >= 2^29 - 1
  __device__ static __forceinline__ uint32_t sub(uint32_t a, uint32_t b, uint32_t kpl) {
    return carry32(a + kpl - norm_exact(b));
  }

  // g * x (mul_by_generator, src/traits.rs:78-91)
  __device__ static __forceinline__ uint32_t mul_g(uint32_t x, const K& k) {
    if constexpr (kTight) return mul(x, k.gm, k.pl);
    return settle_columns((uint64_t)x * (uint32_t)F::kG);
  }

  __device__ static __forceinline__ uint32_t mul_g_settled(uint32_t x, const K& k) { return mul_g(x, k); }   // (coop2d.h's name)

  __device__ static __forceinline__ uint32_t settle(uint32_t x, const K& k) { return mul(x, k.one, k.pl); }

  // x < 2p -> x mod p, exact limbs
  __device__ static __forceinline__ uint32_t canonical(uint32_t x, uint32_t pl) {
    x = norm_exact(x);
    // borrow look-ahead for x - p: generate where x_j < p_j, propagate where equal
    const unsigned long long G = __ballot(x < pl), P = __ballot(x == pl && limb() < NL);
    const unsigned long long B = ((G << 1) + P) ^ P;           // borrow-in per limb
    const bool below = (((G << 1) + P) >> (row0() + NL)) & 1;   // borrow out of this element's top limb: x < p
    const uint32_t d = (x - pl - (uint32_t)((B >> lane()) & 1)) & MASK;
    return below ? x : keep(d);
  }

  // ABI words (lane j < NABI holds 32-bit word j of x * 2^(32 NABI) mod p) -> internal limbs
  __device__ static __forceinline__ uint32_t words_to_limbs(uint32_t w) {
    const int bit = 29 * (int)limb(), lo = bit >> 5, sh = bit & 31, r0 = (int)row0();
    const uint32_t wl = __shfl(w, r0 + (lo < NABI ? lo : 0)), wh = __shfl(w, r0 + (lo + 1 < NABI ? lo + 1 : 0));
    const uint32_t a = lo < NABI ? wl : 0u, b = lo + 1 < NABI ? wh : 0u;
    const uint32_t v = sh == 0 ? a : ((a >> sh) | (b << (32 - sh)));
    return keep(v & MASK);
  }
  __device__ static __forceinline__ uint32_t limbs_to_words(uint32_t l) {  // exact limbs, value < 2^(32 NABI)
    const int bit = 32 * (int)limb(), i0 = bit / 29, off = bit - 29 * i0, r0 = (int)row0();
    const uint32_t l0 = __shfl(l, r0 + (i0 < LPR ? i0 : 0)), l1 = __shfl(l, r0 + (i0 + 1 < LPR ? i0 + 1 : 0)),
                   l2 = __shfl(l, r0 + (i0 + 2 < LPR ? i0 + 2 : 0));
    uint32_t v = l0 >> off;
    v |= off == 0 ? (l1 << 29) : (l1 << (29 - off));
    if (58 - off < 32) v |= l2 << (58 - off);
    return limb() < NABI ? v : 0u;
  }
  __device__ static __forceinline__ uint32_t from_abi(uint32_t w, const K& k) {
    return mul(words_to_limbs(w), k.in, k.pl);
  }
  __device__ static __forceinline__ uint32_t to_abi(uint32_t x, const K& k) {
    return limbs_to_words(canonical(mul(x, k.out, k.pl), k.pl));
  }
};

}  // namespace anemoi
