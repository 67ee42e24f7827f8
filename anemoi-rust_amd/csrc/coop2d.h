// coop2d.h -- the TWO-ROW ("2-D") cooperative Montgomery arithmetic: one field element on a PAIR of 16-lane DPP rows,
// two elements per wavefront.  The lowest-latency path (batches of at most two items per SIMD: the top levels of a
// Merkle tree, a single Jive::compress / Sponge::merge / hash through the shim).
//
// coop29.h's product is a digit-serial scan: NL steps of ten issue slots, each waiting for the quotient digit of the
// step before.  A wavefront alone on its SIMD issues one instruction -- or one wait slot -- per 4.08 cycles whatever the
// dependences (tools/ubench/lone_wave_fetch.hip), so the latency of a compression is the ISSUE-SLOT COUNT of its ~6 500 /
// ~10 000 sequential products (plus what their layout costs: DESIGN 3.5b).  This layout spreads a product
// over two rows and removes the quotient digits altogether (tools/coop2d_model.py is the executable specification):
//
//   layout   NL limbs of W = 28 bits, limb j in lane j of BOTH rows of the pair (lanes >= NL hold zero), Montgomery
//            form with R' = 2^(28 NL), NL chosen so that R'/p >= 2^34 (11 limbs for the 255-bit fields, 15 for the
//            381/377-bit ones): values are only kept below 2^35 p, never below 2p
//   P1       T = a x b, schoolbook, row h multiplying by the limbs a_i with i = h (mod 2): Q = ceil(NL/2) steps of
//            [row broadcast of a_i, two lane shifts of b, two v_mad_u64_u32].  Lane l of LO sums column l - OFF
//            (OFF = 16 - NL), lane l of HI column NL + l: the high half lands where result limb l will live
//   RN1      the low columns are summed over the two rows (v_permlane16_swap) and carried to limbs t_k < 2^28 + 2^5;
//            the carry out of column NL - 1 goes to HI lane 0
//   P2       the low half is folded away by a TABLE instead of the two further products of a Montgomery reduction:
//                T R'^-1  =  TH + sum_k t_k C_k   (mod p),     C_k = 2^(28 k) R'^-1 mod p,
//            C_k[l] per-lane constants (Q registers), the k split over the rows by parity again, accumulated ONTO HI
//   RN2      HI is carried per row, summed over the rows, carried again: limbs < 2^28 + 2^5, value < 2^33 p
//
// ~75 (11 limbs) / ~95 (15 limbs) issue slots per product against ~108 / ~158 for the scan.  The price: results are
// never tight, so (i) values are settled (multiplied by R' mod p) once per round, after the linear layer, (ii) the one product whose
// result has to be < 2p -- the conversion to the ABI form -- runs a digit-serial scan in the same layout (mul_exact),
// once per output element.
//
// Same interface as Coop29 (coop29.h), so the latency kernels of anemoi_coop_kernels.h are instantiated on either.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

#include "field_consts_gen.h"
#include "build_config.h"
#if ANEMOI_COOP2D_NOVDST   // A/B builds: the same statements without hazard padding of DPP destinations (generate the header
#include "coop2d_asm_gen_novdst.h"   // with ANEMOI_COOP2D_GEN_NOVDST=1 python tools/gen_coop2d_asm.py; it is not part of the tree)
#else
#include "coop2d_asm_gen.h"   // tools/gen_coop2d_asm.py: the product below, hand-scheduled
#endif
// ANEMOI_COOP2D_FUSE (build_config.h; 0 in A/B builds: an exponentiation step = a squaring-run statement + a multiplication statement)

namespace anemoi {

// ROWS = 2: an element on a row pair, two per wavefront.  ROWS = 4 (11-limb fields): an element on all four rows, ONE
// per wavefront -- row r multiplies by the limbs a_i, i = r (mod 4), so a phase takes ceil(NL / 4) steps: half the
// multiply-adds, broadcasts and shifts per wavefront for a second swap level (v_permlane32_swap) in the sums over the
// rows.  Built, bit-exact, and MEASURED SLOWER than two rows (18 unfillable hazard slots per product against 6: 86 issue
// slots against 79; Jubjub 1.073 vs 0.960 ms): the recorded negative, compiled into `make AB=1` libraries only (option coop_max).
template <class F, int ROWS = 2>
struct Coop2d {
  using L = typename F::Fold;
  static_assert(ROWS == 2 || (ROWS == 4 && L::Q4 > 0), "four rows: 11-limb fields only (the S form shifts b up by three lanes)");
  static_assert(ROWS == 2 || ANEMOI_AB_BUILD, "the four-row form is a recorded negative: make AB=1");
  static constexpr int W = L::W, NL = L::NL, Q = ROWS == 2 ? L::Q : L::Q4, OFF = L::OFF;
  static constexpr int NABI = F::N;
  static constexpr int kLanesPerItem = 16 * ROWS;
  static constexpr uint32_t MASK = (1u << W) - 1;
  static_assert((W == 27 || W == 28) && NL <= 15 && OFF == 16 - NL && Q == (NL + ROWS - 1) / ROWS, "layout of tools/coop2d_model.py");

  // ---- DPP / cross-row primitives ---------------------------------------------------------------------------------
  template <int CTRL>
  __device__ static __forceinline__ uint32_t dppz(uint32_t v) {  // all rows, out-of-range sources read as 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
  }
  template <int N>
  __device__ static __forceinline__ uint32_t shr(uint32_t v) {  // lane l <- lane l - N
    if constexpr (N == 0) return v;
    else if constexpr (N > 15) return 0u;
    else return dppz<0x110 + N>(v);
  }
  template <int N>
  __device__ static __forceinline__ uint32_t shl(uint32_t v) {  // lane l <- lane l + N
    if constexpr (N == 0) return v;
    else if constexpr (N > 15) return 0u;
    else return dppz<0x100 + N>(v);
  }
  template <int N>
  __device__ static __forceinline__ uint32_t bcast(uint32_t v) {  // lane N of each row to the whole row
    // (bound_ctrl set: every source lane of a row broadcast exists, and without it hipcc materialises the `old`
    // operand with a v_mov and a hazard s_nop in front of every broadcast)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + N, 0xf, 0xf, true);
  }
  __device__ static __forceinline__ uint32_t from_prev(uint32_t v) { return shr<1>(v); }
  // only the odd row of each pair is rewritten (row_mask 0xa); the even row keeps v
  __device__ static __forceinline__ uint32_t odd_rows_from_next(uint32_t v) {  // "D form": odd row lane l holds limb l+1
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x101, 0xa, 0xf, true);
  }
  __device__ static __forceinline__ uint32_t odd_rows_from_prev(uint32_t v) {  // "S form": odd row lane l holds limb l-1
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xa, 0xf, true);
  }
  // v_permlane16_swap: the odd rows of `a` change places with the even rows of `b`
  __device__ static __forceinline__ void swap16(uint32_t& a, uint32_t& b) {
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
  }
  // sum over the two rows of a pair, in both rows
  __device__ static __forceinline__ uint32_t pair_sum(uint32_t v) {
    uint32_t x = v;
    swap16(v, x);  // v = [even, even], x = [odd, odd]
    return v + x;
  }

  __device__ static __forceinline__ uint32_t lane() { return threadIdx.x; }
  __device__ static __forceinline__ uint32_t limb() { return threadIdx.x & 15u; }
  __device__ static __forceinline__ uint32_t row0() { return threadIdx.x & ~15u; }   // first lane of this lane's row
  __device__ static __forceinline__ bool odd_row() { return (threadIdx.x >> 4) & 1u; }
  __device__ static __forceinline__ uint32_t row_in_item() { return (threadIdx.x >> 4) & uint32_t(ROWS - 1); }
  __device__ static __forceinline__ bool writer() { return row_in_item() == 0; }     // the row that stores results
  __device__ static __forceinline__ uint32_t keep(uint32_t v) { return limb() < NL ? v : 0u; }
  __device__ static __forceinline__ uint32_t konst(const uint32_t* __restrict__ k) {
    return limb() < NL ? k[limb()] : 0u;
  }

  struct K {
    uint32_t pl, kpl, delta, one, gm, in, out, rr;
    uint32_t ct[Q];   // fold table: row h, lane l, step q holds limb l of C_(2q+h)
    uint32_t mtop, only15;   // 15 limbs: the limb mask with lane 15 all ones; all ones in lane 15 only
  };
  __device__ static __forceinline__ K load_consts() {
    K k{konst(L::P), konst(L::KP), konst(L::Delta), konst(L::One), konst(L::GMont), konst(L::In), konst(L::Out),
        konst(L::RR), {}};
    const uint32_t h = row_in_item();
#pragma unroll
    for (int q = 0; q < Q; q++) k.ct[q] = ROWS == 2 ? L::FoldT[(q * 2 + h) * 16 + limb()] : L::FoldT4[(q * 4 + h) * 16 + limb()];
    k.mtop = limb() == 15 ? 0xffffffffu : MASK;
    k.only15 = limb() == 15 ? 0xffffffffu : 0u;
    return k;
  }

  // one carry pass for sums of a few almost-normalised limbs (values < 2^32)
  __device__ static __forceinline__ uint32_t carry32(uint32_t r) { return keep((r & MASK) + from_prev(r >> W)); }

  // limbs < 2^29 -> limbs < 2^28 exactly: carry look-ahead on two wavefront ballots (coop29.h); lanes >= NL of every
  // row are zero, neither generate nor propagate, so no carry crosses a row
  __device__ static __forceinline__ uint32_t norm_exact(uint32_t r) {
    r = carry32(r);
    const unsigned long long G = __ballot(r > MASK), P = __ballot(r == MASK);
    const unsigned long long C = ((G << 1) + P) ^ P;
    return (r + (uint32_t)((C >> lane()) & 1)) & MASK;
  }

  // per row: 64-bit column sums -> limbs < 2^W + 2^5 (two carry passes); `hi` and `wh` are handed back for the caller
  // that needs what the top lane shifts out
  __device__ static __forceinline__ uint32_t row_carry(uint64_t acc, uint32_t& hi, uint32_t& wh) {
    const uint32_t lo = (uint32_t)acc & MASK;
    hi = (uint32_t)(acc >> W);
    const uint32_t w = lo + from_prev(hi);
    wh = w >> W;
    return (w & MASK) + from_prev(wh);
  }

  template <int I, class Fn>
  __device__ static __forceinline__ void static_for(Fn&& fn) {   // fn(integral_constant<q>) for q = I .. Q-1
    if constexpr (I < Q) {
      fn(std::integral_constant<int, I>{});
      static_for<I + 1>(fn);
    }
  }

  // a * b * R'^-1 mod p, lazily: result < (A B / H + NL 2^W) p for a < A p, b < B p; limbs < 2^W + 2^5.
  // tools/coop2d_model.py::mul is this function lane for lane.
  __device__ static __forceinline__ uint32_t mul(uint32_t a, uint32_t b, const K& k) {
#if ANEMOI_ASM_MUL
    if constexpr (NL > 13) return AsmCoop2d<NL, W, ROWS>::mul(a, b, k.ct, k.mtop, k.only15);
    else return AsmCoop2d<NL, W, ROWS>::mul(a, b, k.ct);
#endif
    return mul_cxx(a, b, k);
  }
  // x^(2^n), n >= 1: a run of squarings (the assembly keeps the loop inside one statement)
  __device__ static __forceinline__ uint32_t sqr_n(uint32_t a, uint32_t n, const K& k) {
#if ANEMOI_ASM_MUL
    if constexpr (NL > 13) return AsmCoop2d<NL, W, ROWS>::sqr_run(a, k.ct, k.mtop, k.only15, n);
    else return AsmCoop2d<NL, W, ROWS>::sqr_run(a, k.ct, n);
#endif
    for (uint32_t i = 0; i < n; i++) a = mul_cxx(a, a, k);
    return a;
  }
  // x^(2^n) * b, n >= 1: one step of the sliding-window exponentiation (the assembly issues b's lane shifts in the
  // hazard gaps of the last squaring)
  // Measured in one process (profiles/r04/ab_coop2d_fused_steps.txt): the fused statement is 3.6 % faster on 11 limbs
  // (Jubjub 1.118 -> 1.078 ms) and 9 % SLOWER on 15 (BLS12-381 1.838 -> 2.010 ms: sixteen more live registers and
  // eighteen prefetched shifts for a step that is mostly squarings), so the 15-limb fields keep two statements per step.
  static constexpr bool kFuseSteps = ANEMOI_COOP2D_FUSE && NL <= 13;   // (both row counts of the 11-limb fields)
  __device__ static __forceinline__ uint32_t sqr_mul(uint32_t a, uint32_t n, uint32_t b, const K& k) {
#if ANEMOI_ASM_MUL
    if constexpr (!kFuseSteps) return mul(sqr_n(a, n, k), b, k);
    else if constexpr (NL > 13) return AsmCoop2d<NL, W, ROWS>::sqr_mul(a, b, k.ct, k.mtop, k.only15, n);
    else return AsmCoop2d<NL, W, ROWS>::sqr_mul(a, b, k.ct, n);
#endif
    for (uint32_t i = 0; i < n; i++) a = mul_cxx(a, a, k);
    return mul_cxx(a, b, k);
  }
  // the readable form of the same product (ANEMOI_ASM_MUL=0 builds run it)
  __device__ static __forceinline__ uint32_t mul_cxx(uint32_t a, uint32_t b, const K& k) {
#if ANEMOI_AB_BUILD
    if constexpr (ROWS == 4) return mul_cxx4(a, b, k);
#endif
    const uint32_t aD = odd_rows_from_next(a), bS = odd_rows_from_prev(b);
    uint64_t LO = 0, HI = 0;
    static_for<0>([&](auto I) {   // P1
      constexpr int q = decltype(I)::value;
      const uint32_t aq = bcast<2 * q>(aD);
      LO += (uint64_t)aq * shr<OFF + 2 * q>(bS);
      HI += (uint64_t)aq * shl<NL - 2 * q>(bS);
    });
    // RN1: low columns (LO lanes OFF..15) -> limbs t; what leaves the top lane is the carry into column NL = HI lane 0
    uint32_t t;
    if constexpr (NL <= 13) {   // sum over the rows first: sum of the high parts <= NL 2^W, v < (NL + 2) 2^W
      uint32_t lo = (uint32_t)LO & MASK, hi = (uint32_t)(LO >> W);
      swap16(lo, hi);
      uint32_t s = lo + hi;     // even row: sum of the low parts; odd row: sum of the high parts
      uint32_t x = s;
      swap16(s, x);             // s = low parts, x = high parts, in both rows
      const uint32_t v = s + from_prev(x), vh = v >> W;
      const uint32_t cc = x + vh;   // lane 15: everything column NL-1 hands on; lanes 0..2 are zero (OFF >= 3)
      static_assert(NL > 13 || OFF >= 3, "the carry injection relies on empty lanes 0..2");
      t = (v & MASK) + from_prev(vh);
      // rows 0 and 2 only (the rows are summed in RN2), lane 0 <- lane 15 (row_ror:1, bank 0 = lanes 0..3)
      HI += (uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)cc, 0x121, 0x5, 0x1, false);
    } else {                    // 15 limbs: carry inside each row first, then sum over the rows
      uint32_t hi, wh;
      const uint32_t w2 = row_carry(LO, hi, wh);
      const uint32_t ccr = limb() == 15 ? hi + wh : 0u;   // this row's share of the carry into column NL
      HI += (uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)ccr, 0x121, 0xf, 0x1, false);
      const uint32_t y = pair_sum(w2), yh = y >> W;
      t = (limb() == 15 ? y : (y & MASK)) + from_prev(yh);   // the top limb keeps its own carry (t_15 < 2^29 + 2^6)
    }
    // P2: fold the low half with the table
    const uint32_t tD = odd_rows_from_next(t);
    static_for<0>([&](auto I) {
      constexpr int q = decltype(I)::value;
      HI += (uint64_t)bcast<OFF + 2 * q>(tD) * k.ct[q];
    });
    // RN2
    if constexpr (W <= 27) {   // the 64-bit sum over the rows first: <= 21 products of 2^54 leave a 32-bit carry
      uint32_t l0 = (uint32_t)HI, h0 = (uint32_t)(HI >> 32), l1 = l0, h1 = h0;
      swap16(l0, l1);
      swap16(h0, h1);
      const uint64_t tot = (((uint64_t)h0 << 32) | l0) + (((uint64_t)h1 << 32) | l1);
      const uint32_t w = ((uint32_t)tot & MASK) + from_prev((uint32_t)(tot >> W));
      return (w & MASK) + from_prev(w >> W);
    } else {
      uint32_t hi, wh;
      const uint32_t w2 = row_carry(HI, hi, wh);
      const uint32_t y = pair_sum(w2);
      // (lanes >= NL stay zero by themselves: no product or table entry reaches them and the value is far below R')
      return (y & MASK) + from_prev(y >> W);
    }
  }

#if ANEMOI_AB_BUILD
  // ---- four rows per element (tools/coop2d_model.py::mul4): `make AB=1` libraries only ---------------------------------------
  __device__ static __forceinline__ void swap32(uint32_t& a, uint32_t& b) {   // rows 2, 3 of a <-> rows 0, 1 of b
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
  }
  template <int SHL>   // rows 1, 2, 3 shifted by 1, 2, 3 lanes (down: D form of a multiplier; up: S form of a multiplicand)
  __device__ static __forceinline__ uint32_t skew(uint32_t v) {
    constexpr int base = SHL ? 0x100 : 0x110;
    uint32_t r = v;
    r = (uint32_t)__builtin_amdgcn_update_dpp((int)r, (int)v, base + 1, 0x2, 0xf, true);
    r = (uint32_t)__builtin_amdgcn_update_dpp((int)r, (int)v, base + 2, 0x4, 0xf, true);
    r = (uint32_t)__builtin_amdgcn_update_dpp((int)r, (int)v, base + 3, 0x8, 0xf, true);
    return r;
  }
  __device__ static __forceinline__ uint64_t sum4(uint64_t v) {   // the sum over the four rows, in every row
    uint32_t l0 = (uint32_t)v, h0 = (uint32_t)(v >> 32), l1 = l0, h1 = h0;
    swap16(l0, l1);
    swap16(h0, h1);
    v = (((uint64_t)h0 << 32) | l0) + (((uint64_t)h1 << 32) | l1);
    l0 = (uint32_t)v, h0 = (uint32_t)(v >> 32), l1 = l0, h1 = h0;
    swap32(l0, l1);
    swap32(h0, h1);
    return (((uint64_t)h0 << 32) | l0) + (((uint64_t)h1 << 32) | l1);
  }
  __device__ static __forceinline__ uint32_t mul_cxx4(uint32_t a, uint32_t b, const K& k) {
    static_assert(ROWS != 4 || (NL <= 13 && W <= 27), "the four-row form is built for the 11-limb, 27-bit layout");
    const uint32_t aD = skew<1>(a), bS = skew<0>(b);
    uint64_t LO = 0, HI = 0;
    static_for<0>([&](auto I) {
      constexpr int q = decltype(I)::value;
      const uint32_t aq = bcast<4 * q>(aD);
      LO += (uint64_t)aq * shr<OFF + 4 * q>(bS);
      HI += (uint64_t)aq * shl<NL - 4 * q>(bS);
    });
    // RN1: low and high parts summed over the four rows (swap16, swap32), then distributed (swap16)
    uint32_t lo = (uint32_t)LO & MASK, hi = (uint32_t)(LO >> W);
    swap16(lo, hi);
    uint32_t s = lo + hi, x = s;          // rows: lo0+lo1, hi0+hi1, lo2+lo3, hi2+hi3
    swap32(s, x);
    uint32_t u = s + x, y = u;            // rows: low parts, high parts, low parts, high parts
    swap16(u, y);                         // u = low parts, y = high parts, in all four rows
    const uint32_t v = u + from_prev(y), vh = v >> W;
    const uint32_t cc = y + vh;           // lane 15: everything column NL-1 hands on; lanes 0..2 are zero (OFF >= 3)
    const uint32_t t = (v & MASK) + from_prev(vh);
    HI += (uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)cc, 0x121, 0x1, 0x1, false);   // row 0, lane 0 <- lane 15
    const uint32_t tD = skew<1>(t);
    static_for<0>([&](auto I) {           // P2: fold
      constexpr int q = decltype(I)::value;
      HI += (uint64_t)bcast<OFF + 4 * q>(tD) * k.ct[q];
    });
    const uint64_t tot = sum4(HI);        // RN2: the 64-bit sum over the rows, two carry passes
    const uint32_t w = ((uint32_t)tot & MASK) + from_prev((uint32_t)(tot >> W));
    return (w & MASK) + from_prev(w >> W);
  }
#endif  // ANEMOI_AB_BUILD

  // Digit-serial Montgomery product (the scan of coop29.h on this layout, both rows doing the same work): result
  // < (A B / H + 1) p.  Only where a result has to be < 2p: the conversion to the ABI form.
  template <int I, class Fn>
  __device__ static __forceinline__ void static_for_limbs(Fn&& fn) {
    if constexpr (I < NL) {
      fn(std::integral_constant<int, I>{});
      static_for_limbs<I + 1>(fn);
    }
  }
  __device__ static __forceinline__ uint32_t mul_exact(uint32_t a, uint32_t b, uint32_t pl) {
    const uint32_t sh = limb() == 0 ? uint32_t(W) : 63u;
    uint64_t t = 0;
    static_for_limbs<0>([&](auto I) {
      constexpr int i = decltype(I)::value;
      t += (uint64_t)bcast<i>(a) * b;
      const uint32_t m = (bcast<0>((uint32_t)t) * L::kN0Inv) & MASK;
      t += (uint64_t)m * pl;
      const uint64_t u = t >> sh;   // lane 0: the retired column's carry; elsewhere 0 (column sums < 2^63)
      t = (((uint64_t)shl<1>((uint32_t)(t >> 32)) << 32) | shl<1>((uint32_t)t)) + u;
    });
    const uint32_t lo = (uint32_t)t & MASK;
    const uint64_t hi = t >> W;
    const uint64_t up = ((uint64_t)from_prev((uint32_t)(hi >> 32)) << 32) | from_prev((uint32_t)hi);
    const uint64_t r = (uint64_t)lo + up;
    return keep(((uint32_t)r & MASK) + from_prev((uint32_t)(r >> W)));
  }

  __device__ static __forceinline__ uint32_t add(uint32_t a, uint32_t b) { return carry32(a + b); }
  __device__ static __forceinline__ uint32_t settle(uint32_t x, const K& k) { return mul(x, k.one, k); }

  // a - b + 2^s p for b < 2^s p (what the S-box subtracts: g * (a product of settled values) or a product; the pad is
  // field_consts_gen.h's KP).  NOT settled: R'/p >= 2^39 leaves room for the few pads a round accumulates before
  // coop_permutation settles x and y after the linear layer -- tests/test_coop2d_model.py walks the bounds of a round
  // for every field (every subtrahend <= the pad, every value < R', every product's column sums inside 64 bits).
  __device__ static __forceinline__ uint32_t sub(uint32_t a, uint32_t b, const K& k) {
    return carry32(a + k.kpl - norm_exact(b));
  }

  // g * x (mul_by_generator, src/traits.rs:78-91)
  __device__ static __forceinline__ uint32_t mul_g(uint32_t x, const K& k) {
    if constexpr (L::kScaleG) return carry32(x * (uint32_t)F::kG);
    else return mul(x, k.gm, k);
  }

  // g * x as a PRODUCT (the result is as small as any product's): the 4-3 linear layer, whose sums would not stay below R'
  __device__ static __forceinline__ uint32_t mul_g_settled(uint32_t x, const K& k) { return mul(x, k.gm, k); }

  // x < 2p -> x mod p, exact limbs (borrow look-ahead, as coop29.h)
  __device__ static __forceinline__ uint32_t canonical(uint32_t x, uint32_t pl) {
    x = norm_exact(x);
    const unsigned long long G = __ballot(x < pl), P = __ballot(x == pl && limb() < NL);
    const unsigned long long B = ((G << 1) + P) ^ P;
    const bool below = (((G << 1) + P) >> (row0() + NL)) & 1;
    const uint32_t d = (x - pl - (uint32_t)((B >> lane()) & 1)) & MASK;
    return below ? x : keep(d);
  }

  // ABI words (lane j < NABI holds 32-bit word j) <-> limbs
  __device__ static __forceinline__ uint32_t words_to_limbs(uint32_t w) {
    const int bit = W * (int)limb(), lo = bit >> 5, sh = bit & 31, r0 = (int)row0();
    const uint32_t wl = __shfl(w, r0 + (lo < NABI ? lo : 0)), wh = __shfl(w, r0 + (lo + 1 < NABI ? lo + 1 : 0));
    const uint32_t a = lo < NABI ? wl : 0u, b = lo + 1 < NABI ? wh : 0u;
    const uint32_t v = sh == 0 ? a : ((a >> sh) | (b << (32 - sh)));
    return keep(v & MASK);
  }
  __device__ static __forceinline__ uint32_t limbs_to_words(uint32_t l) {  // exact limbs, value < 2^(32 NABI)
    const int bit = 32 * (int)limb(), i0 = bit / W, off = bit - W * i0, r0 = (int)row0();
    const uint32_t l0 = __shfl(l, r0 + (i0 < 16 ? i0 : 0)), l1 = __shfl(l, r0 + (i0 + 1 < 16 ? i0 + 1 : 0)),
                   l2 = __shfl(l, r0 + (i0 + 2 < 16 ? i0 + 2 : 0));
    uint32_t v = i0 < 16 ? l0 >> off : 0u;
    if (i0 + 1 < 16) v |= l1 << (W - off);
    if (2 * W - off < 32 && i0 + 2 < 16) v |= l2 << (2 * W - off);
    return limb() < NABI ? v : 0u;
  }
  __device__ static __forceinline__ uint32_t from_abi(uint32_t w, const K& k) { return mul(words_to_limbs(w), k.in, k); }
  __device__ static __forceinline__ uint32_t to_abi(uint32_t x, const K& k) {
    return limbs_to_words(canonical(mul_exact(x, k.out, k.pl), k.pl));
  }
  // plain integer < 2^(W NL) (limbs) -> Montgomery form
  __device__ static __forceinline__ uint32_t to_mont(uint32_t x, const K& k) { return mul(x, k.rr, k); }
};

}  // namespace anemoi
