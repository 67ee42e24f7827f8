// mont29.h -- lane-private Montgomery arithmetic on unsaturated 29-bit limbs for gfx950 (CDNA4).
//
// Why not 32-bit limbs: measured on MI355X (tools/ubench/wall_rates.hip, profiles/r01/), a
// v_mad_u64_u32 (4.6-5 cycles per wave-instruction per SIMD) costs about the same as a
// v_addc_co_u32 / v_lshl_add_u64 (4.2-4.4), and a VALU-written carry (VCC/SGPR) needs 2 wait states
// before a VALU can consume it.  With saturated 32-bit limbs every multiply-accumulate therefore
// costs a multiply PLUS a carry instruction.  With 29-bit limbs a whole column of the product
// (<= 28 products < 2^58) fits a 64-bit accumulator, so a multiply-accumulate is exactly ONE
// v_mad_u64_u32 and carries are resolved once per column (one shift, one mask):
//     BLS12-381 squaring: 301 multiplies + 111 cheap ops  vs  234 multiplies + 234 carries + pads.
// The spare bits also remove every conditional subtraction: with H = R'/p (R' = 2^(29*NL); 2^25 for
// the 381/377-bit fields, 70..438 for the 253..255-bit ones) a Montgomery product of inputs < A p
// and < B p is < (A B / H + 1) p, so additions / subtractions need no reduction at all and values
// are made canonical once, on the way out.  The bounds every statement of every kernel meets are walked on the
// kernels' own code and listed in BOUNDS.md (tests/cpp/bounds_walk, tools/bounds_walk.py, tests/test_bounds_walk.py).
//
// Element form inside the kernels: NL limbs l[i] < 2^29, value = sum l[i] 2^(29 i), Montgomery form
// with R' = 2^(29 NL).  The C-ABI form (arkworks: 32-bit-limb R = 2^(64 L)) is converted on load
// and store by one Montgomery product each (from_abi / to_abi), exact and bit-identical.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "build_config.h"
// ANEMOI_ASM_MUL = 1 (the product): squaring / multiplication come from the generated hand-scheduled assembly
// (mont29_asm_gen.h, tools/gen_asm_mul.py); 0 (A/B builds, the host walk of the bounds): the C++ below, which is the
// readable specification of the same arithmetic.
#if ANEMOI_ASM_MUL
#include "mont29_asm_gen.h"
#endif

namespace anemoi {

template <class F, class L = typename F::Lane>
struct Arith29 {
  static constexpr int W = L::W;       // bits per limb: 29, or 30 for the 381/377-bit fields (see below)
  static constexpr int NL = L::NL;     // limbs held in VGPRs
  static constexpr int NABI = F::N;    // 32-bit words of an ABI element
  static constexpr int NQ = (NL + 3) / 4;  // uint4 slots of one LDS table entry
  static constexpr uint32_t MASK = (1u << W) - 1;
  static constexpr bool kLoose = true;  // values between reductions may exceed p (see settle())

  struct Fe {
    uint32_t l[NL];
  };

  // A field constant as the `b` operand of a product.  The limbs are moved into VGPRs by volatile statements
  // right where they are needed: written as plain assignments, hipcc hoists the 13 v_mov of `One` (settle) and of
  // `g R'` (mul_g of the tight fields) out of the round loop and keeps 26 registers alive across the whole
  // exponentiation -- that alone put the BLS12-377 kernels at 180 VGPRs = 2 waves per SIMD
  // (tools/vgpr_liveness.py, profiles/r03/kernel_resources.csv).  13 v_mov per use is noise next to ~430
  // instructions per product.
  template <const uint32_t (&K)[NL], int I = 0>
  __device__ static __forceinline__ void load_const(Fe& k) {
    if constexpr (I < NL) {
      asm volatile("v_mov_b32 %0, %1" : "=v"(k.l[I]) : "n"(K[I]));
      load_const<K, I + 1>(k);
    }
  }

  // carry ripple: limbs back below 2^29 (the top limb keeps the excess; values stay < 2^(29 NL))
  __device__ static __forceinline__ void norm(Fe& r) {
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
      r.l[i + 1] += r.l[i] >> W;
      r.l[i] &= MASK;
    }
  }

  // Montgomery product a*b/R' mod p, result < 2p when (a/p)*(b/p) <= R'/p.  All limbs < 2^W.
  // On 30-bit limbs the generated assembly additionally assumes b < 16 p (mul) and a < 16 p (sqr): that
  // bounds the top limb and saves column splits (tools/gen_asm_mul.py).  Every call site satisfies it: b
  // is a table entry, a fresh product, a constant or the S-box input (< 15 p); squarings only see S-box
  // values (< 15 p) and chain values (< 2 p).  Values up to H p (the linear layer) only enter as `a` of mul.
  //
  // Product scanning, column k:  acc_k = (acc_{k-1} >> 29) + X_k + M_k + m_k p_0  with
  //   X_k = sum_j a_j b_{k-j}            (no dependence on the reduction digits m)
  //   M_k = sum_{j<k} m_j p_{k-j}        (only its last term m_{k-1} p_1 waits for the previous column)
  //   m_k = -(low 29 bits) / p mod 2^29
  // hipcc reassociates these sums into an operand-scanning schedule with ~20 live column
  // accumulators (150 VGPRs, 3 waves/SIMD): 301 v_mad_u64_u32 + 111 other instructions per squaring.
  // Forcing a leaner schedule (fewer registers, 5-8 waves/SIMD) was measured and is slower: the
  // multiplier is bound by VALU issue, not by occupancy (profiles/r01/ab_occupancy_variants.txt).
  __device__ static __forceinline__ void mul(Fe& r, const Fe& a, const Fe& b) {
#if ANEMOI_ASM_MUL
    Fe t = a;
    AsmMont<F::kId, W>::mul(t.l, b.l);
    r = t;
    return;
#endif
    uint32_t m[NL], out[NL];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
      const int j0 = k < NL ? 0 : k - NL + 1, j1 = k < NL ? k : NL - 1;
      uint64_t x = 0;
#pragma unroll
      for (int j = j0; j <= j1; j++) x += (uint64_t)a.l[j] * b.l[k - j];
      column_tail(acc, x, m, out, k);
    }
    out[NL - 1] = (uint32_t)acc;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = out[i];
  }

  // Montgomery square.  Off-diagonal products use a pre-doubled copy of a (limb-wise doubling is
  // exact with unsaturated limbs: 2*l[i] < 2^30), so each costs one multiply-accumulate.
  __device__ static __forceinline__ void sqr(Fe& r, const Fe& a) {
#if ANEMOI_ASM_MUL
    Fe t = a;
    AsmMont<F::kId, W>::sqr(t.l);
    r = t;
    return;
#endif
    uint32_t m[NL], out[NL], a2[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) a2[i] = a.l[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
      const int j0 = k < NL ? 0 : k - NL + 1;
      uint64_t x = 0;
#pragma unroll
      for (int j = j0; j < k - j; j++) x += (uint64_t)a2[j] * a.l[k - j];
      if ((k & 1) == 0) x += (uint64_t)a.l[k / 2] * a.l[k / 2];
      column_tail(acc, x, m, out, k);
    }
    out[NL - 1] = (uint32_t)acc;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = out[i];
  }

  // reduction half of column k (shared by mul and sqr); x = the column's product sum
  __device__ static __forceinline__ void column_tail(uint64_t& acc, uint64_t x, uint32_t (&m)[NL], uint32_t (&out)[NL],
                                                     const int k) {
    const int j0 = k < NL ? 0 : k - NL + 1, j1 = k < NL ? k - 1 : NL - 1;  // m_j p_{k-j}, j in [j0, j1]
    uint64_t early = x;
#pragma unroll
    for (int j = j0; j < j1; j++) early += (uint64_t)m[j] * L::P[k - j];
    if constexpr (W >= 30) {
      // 30-bit limbs: a column (26 products < 2^60 each) no longer fits 64 bits -- the generated
      // assembly splits the heavy columns; this C++ form simply carries 128 bits
      unsigned __int128 wide = (unsigned __int128)acc + early;
      if (j1 >= j0) wide += (uint64_t)m[j1] * L::P[k - j1];
      if (k < NL) {
        m[k] = ((uint32_t)wide * L::kN0Inv) & MASK;
        wide += (uint64_t)m[k] * L::P[0];
      } else {
        out[k - NL] = (uint32_t)wide & MASK;
      }
      acc = (uint64_t)(wide >> W);
      return;
    }
    acc += early;
    if (j1 >= j0) acc += (uint64_t)m[j1] * L::P[k - j1];
    if (k < NL) {
      m[k] = ((uint32_t)acc * L::kN0Inv) & MASK;
      acc += (uint64_t)m[k] * L::P[0];
    } else {
      out[k - NL] = (uint32_t)acc & MASK;
    }
    acc >>= W;
  }

  // r = a + b (no reduction; the sum must stay below R' = 2^(W NL): the walk of tests/test_bounds_walk.py checks every call)
  __device__ static __forceinline__ void add(Fe& r, const Fe& a, const Fe& b) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i];
    norm(r);
  }

  // r = a + k, k given as NL limbs (compile-time or wave-uniform)
  __device__ static __forceinline__ void add_k(Fe& r, const Fe& a, const uint32_t* __restrict__ k) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + k[i];
    norm(r);
  }

  // r = a - b + kSubK p >= 0: KP is kSubK p with its lower limbs padded to >= 2^W - 1 (tools/gen_params.py), so no limb
  // goes negative while b's TOP limb stays <= KP's (b < kSubK p, a hair less); every call is checked by the bounds walk
  __device__ static __forceinline__ void sub(Fe& r, const Fe& a, const Fe& b) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + L::KP[i] - b.l[i];
    norm(r);
  }

  // Fields with little headroom (R'/p < 2^12: the 254/255-bit ones on 9 limbs) keep every value
  // below ~33 p: there g*x is a Montgomery product by g*R' (result < 2p) and subtraction pads with 4p.
  static constexpr bool kTight = L::kTight;

  // r = g * x for the small generator g (mul_by_generator, src/traits.rs:78-91)
  __device__ static __forceinline__ void mul_g(Fe& r, const Fe& x) {
    if constexpr (kTight) {
      Fe g;
      load_const<L::GMont>(g);
      mul(r, x, g);
    } else {
      uint64_t acc = 0;
#pragma unroll
      for (int i = 0; i < NL; i++) {
        acc += (uint64_t)x.l[i] * (uint32_t)F::kG;
        r.l[i] = (uint32_t)acc & MASK;
        acc >>= W;
      }
    }
  }

  // value-preserving reduction to < 2p: x * R' / R'
  __device__ static __forceinline__ void settle(Fe& x) {
    Fe one;
    load_const<L::One>(one);
    mul(x, x, one);
  }

  __device__ static __forceinline__ void set_one(Fe& x) {
#pragma unroll
    for (int i = 0; i < NL; i++) x.l[i] = L::One[i];
  }
  __device__ static __forceinline__ void set_zero(Fe& x) {
#pragma unroll
    for (int i = 0; i < NL; i++) x.l[i] = 0;
  }
  __device__ static __forceinline__ void add_delta(Fe& r, const Fe& a) { add_k(r, a, L::Delta); }

  // x < 2p (normalised) -> x mod p
  __device__ static __forceinline__ void canonical(Fe& x) {
    uint32_t d[NL];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      uint32_t t = x.l[i] - L::P[i] - borrow;
      borrow = t >> 31;
      d[i] = t & MASK;
    }
#pragma unroll
    for (int i = 0; i < NL; i++) x.l[i] = borrow ? x.l[i] : d[i];
  }

  // 32-bit words (plain integer) -> 29-bit limbs (same integer).  32-bit funnel shifts only
  // (v_alignbit_b32): a 64-bit formulation makes hipcc build the value through scratch memory.
  __device__ static __forceinline__ void repack_in(Fe& r, const uint32_t (&w)[NABI]) {
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int bit = W * i, lo = bit >> 5, sh = bit & 31;
      const uint32_t wl = lo < NABI ? w[lo] : 0u;
      const uint32_t wh = lo + 1 < NABI ? w[lo + 1] : 0u;
      const uint32_t v = sh == 0 ? wl : ((wl >> sh) | (wh << (32 - sh)));
      r.l[i] = v & MASK;
    }
  }
  // 29-bit limbs (value < 2^(32 NABI)) -> 32-bit words
  __device__ static __forceinline__ void repack_out(uint32_t (&w)[NABI], const Fe& a) {
#pragma unroll
    for (int j = 0; j < NABI; j++) {
      const int bit = 32 * j, i0 = bit / W, off = bit - W * i0;
      uint32_t v = a.l[i0] >> off;                                  // W - off bits
      if (i0 + 1 < NL) v |= a.l[i0 + 1] << (W - off);               // next W bits
      if (i0 + 2 < NL && 2 * W - off < 32) v |= a.l[i0 + 2] << (2 * W - off);
      w[j] = v;
    }
  }

  // ABI element (x * 2^(32 NABI) mod p, 32-bit words) -> internal (x * R' mod p), < 2p
  __device__ static __forceinline__ void from_abi(Fe& r, const uint32_t (&w)[NABI]) {
    Fe k;
    repack_in(r, w);
    load_const<L::In>(k);
    mul(r, r, k);
  }
  // internal (any value < 2^12 p) -> canonical ABI element
  __device__ static __forceinline__ void to_abi(uint32_t (&w)[NABI], const Fe& a) {
    Fe k, t;
    load_const<L::Out>(k);
    mul(t, a, k);
    canonical(t);
    repack_out(w, t);
  }
  // plain integer < p (32-bit words) -> internal Montgomery form
  __device__ static __forceinline__ void from_int(Fe& r, const uint32_t (&w)[NABI]) {
    Fe k;
    repack_in(r, w);
    load_const<L::RR>(k);
    mul(r, r, k);
  }

  // exponentiation-chain forms (same as mul/sqr here: results are always < 2p)
  __device__ static __forceinline__ void emul(Fe& r, const Fe& a, const Fe& b) { mul(r, a, b); }
  __device__ static __forceinline__ void esqr(Fe& r, const Fe& a) { sqr(r, a); }
  __device__ static __forceinline__ void efinish(Fe&) {}

  // LDS table entry <-> registers (NQ uint4 slots, `stride` uint4 apart)
  __device__ static __forceinline__ void lds_store(uint4* base, int stride, const Fe& v) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      uint4 t;
      t.x = v.l[4 * q];
      t.y = 4 * q + 1 < NL ? v.l[4 * q + 1] : 0u;
      t.z = 4 * q + 2 < NL ? v.l[4 * q + 2] : 0u;
      t.w = 4 * q + 3 < NL ? v.l[4 * q + 3] : 0u;
      base[q * stride] = t;
    }
  }
  __device__ static __forceinline__ void lds_load(const uint4* base, int stride, Fe& v) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      uint4 t = base[q * stride];
      v.l[4 * q] = t.x;
      if (4 * q + 1 < NL) v.l[4 * q + 1] = t.y;
      if (4 * q + 2 < NL) v.l[4 * q + 2] = t.z;
      if (4 * q + 3 < NL) v.l[4 * q + 3] = t.w;
    }
  }

  static const uint32_t* host_ark(int width, bool d) {
    return width == 2 ? (d ? L::ArkD_21 : L::ArkC_21) : (d ? L::ArkD_43 : L::ArkC_43);
  }
};

}  // namespace anemoi
