// mont32.h -- lane-private Montgomery arithmetic on N x 32-bit limbs for gfx950 (CDNA4).
//
// Replaces arkworks' `Fp<MontBackend<_, N>, N>` (the reference's Felt, src/<field>/mod.rs:1-3,
// Cargo.toml:15-22) inside the kernels.  Same value representation: little-endian limbs,
// Montgomery form with R = 2^(32*N) = 2^(64*limbs64), so device buffers are byte-identical to a
// Rust `&[Felt]`.
//
// One field element per lane, limbs in VGPRs, every loop fully unrolled.  The multiplier is a
// finely-integrated product-scanning (FIPS) Montgomery product: each 32-bit column is summed in a
// 96-bit accumulator (c0 + 2^32*c12) with
//     v_mad_u64_u32   t[1:0], a, b, {c0,0}              ; 32x32 + 32 -> 64, cannot overflow
//     v_lshl_add_u64  c12, {t1,0}, 0, c12               ; fold the high word
// The modulus limbs are compile-time constants and reach the multiplier through SGPRs, so they cost
// no VGPRs.  Squaring sums the off-diagonal products once and doubles the partial column.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "build_config.h"

namespace anemoi {

template <int N>
struct Fe {
  uint32_t l[N];
};

// ANEMOI_MAC_MODE (build_config.h): 1 = the shipped multiply-accumulate (asm v_mad_u64_u32 carry-out + deferred v_addc),
// 0 = plain C++ (A/B builds, and the host compile of the bounds walk)

#if ANEMOI_MAC_MODE == 0
// ---- mode 0: plain C++ ------------------------------------------------------------------------
// 96-bit column accumulator: value = c0 + 2^32 * c12.
struct Acc {
  uint32_t c0;
  uint64_t c12;
};

// acc += a * b.  a*b + c0 < 2^64, so the 32x32+64 multiply-add (v_mad_u64_u32) cannot overflow and
// no carry flag is needed; the high word is folded into c12 with one 64-bit add (v_lshl_add_u64).
__device__ __forceinline__ void mac(Acc& acc, uint32_t a, uint32_t b) {
  uint64_t t = (uint64_t)a * b + acc.c0;
  acc.c0 = (uint32_t)t;
  acc.c12 += t >> 32;
}
__device__ __forceinline__ void mac_k(Acc& acc, uint32_t a, uint32_t k) { mac(acc, a, k); }

// acc >>= 32
__device__ __forceinline__ void acc_shift(Acc& acc) {
  acc.c0 = (uint32_t)acc.c12;
  acc.c12 >>= 32;
}
__device__ __forceinline__ uint32_t acc_lo(const Acc& acc) { return acc.c0; }
// acc += 2 * x
__device__ __forceinline__ void acc_add2(Acc& acc, const Acc& x) {
  uint64_t s = (uint64_t)acc.c0 + ((uint64_t)x.c0 << 1);
  acc.c0 = (uint32_t)s;
  acc.c12 += (x.c12 << 1) + (s >> 32);
}

// acc += sum_{j=J0}^{J0+CNT-1} a[j] * b[K-j]
template <int K, int J0, int CNT, int N>
__device__ __forceinline__ void mac_run_vv(Acc& acc, const uint32_t (&a)[N], const uint32_t (&b)[N]) {
#pragma unroll
  for (int j = J0; j < J0 + CNT; j++) mac(acc, a[j], b[K - j]);
}
// acc += sum_{j=J0}^{J0+CNT-1} m[j] * P[K-j]
template <class F, int K, int J0, int CNT, int N>
__device__ __forceinline__ void mac_run_vp(Acc& acc, const uint32_t (&m)[N]) {
#pragma unroll
  for (int j = J0; j < J0 + CNT; j++) mac_k(acc, m[j], F::P[K - j]);
}

#else
// ---- mode 1: v_mad_u64_u32 carry-out + deferred v_addc -------------------------------------------
// 96-bit column accumulator: value = lo + 2^64 * hi.  Each multiply-add is
//     v_mad_u64_u32  lo, s[c], a, b, lo      ; 32x32 + 64 -> 64, carry-out mask into an SGPR pair
//     v_addc_co_u32  hi, vcc, 0, hi, s[c]    ; fold the carry
// gfx950 needs 2 wait states between a VALU write of an SGPR and a VALU read of it as carry-in
// (and hipcc pads nothing inside an asm statement), so MACs are issued in blocks of 4 / 2 / 1 with
// the v_addc's deferred behind the block's multiplies (or behind an s_nop).
struct Acc {
  uint64_t lo;
  uint32_t hi;
};

#define ANEMOI_MAC4(CB)                                                                             \
  uint64_t s0, s1, s2, s3;                                                                          \
  asm("v_mad_u64_u32 %0, %2, %6, %7, %0\n\t"                                                        \
      "v_mad_u64_u32 %0, %3, %8, %9, %0\n\t"                                                        \
      "v_mad_u64_u32 %0, %4, %10, %11, %0\n\t"                                                      \
      "v_mad_u64_u32 %0, %5, %12, %13, %0\n\t"                                                      \
      "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"                                                        \
      "v_addc_co_u32 %1, vcc, 0, %1, %3\n\t"                                                        \
      "v_addc_co_u32 %1, vcc, 0, %1, %4\n\t"                                                        \
      "v_addc_co_u32 %1, vcc, 0, %1, %5"                                                            \
      : "+v"(acc.lo), "+v"(acc.hi), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3)                      \
      : "v"(a0), CB(b0), "v"(a1), CB(b1), "v"(a2), CB(b2), "v"(a3), CB(b3)                          \
      : "vcc")
#define ANEMOI_MAC2(CB)                                                                             \
  uint64_t s0, s1;                                                                                  \
  asm("v_mad_u64_u32 %0, %2, %4, %5, %0\n\t"                                                        \
      "v_mad_u64_u32 %0, %3, %6, %7, %0\n\t"                                                        \
      "s_nop 0\n\t"                                                                                 \
      "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"                                                        \
      "v_addc_co_u32 %1, vcc, 0, %1, %3"                                                            \
      : "+v"(acc.lo), "+v"(acc.hi), "=&s"(s0), "=&s"(s1)                                            \
      : "v"(a0), CB(b0), "v"(a1), CB(b1)                                                            \
      : "vcc")
#define ANEMOI_MAC1(CB)                                                                             \
  uint64_t s0;                                                                                      \
  asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\t"                                                        \
      "s_nop 1\n\t"                                                                                 \
      "v_addc_co_u32 %1, vcc, 0, %1, %2"                                                            \
      : "+v"(acc.lo), "+v"(acc.hi), "=&s"(s0)                                                       \
      : "v"(a0), CB(b0)                                                                             \
      : "vcc")
#define ANEMOI_CV(x) "v"(x)
#define ANEMOI_CS(x) "s"(x)

__device__ __forceinline__ void mac4_vv(Acc& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2,
                                        uint32_t b2, uint32_t a3, uint32_t b3) {
  ANEMOI_MAC4(ANEMOI_CV);
}
__device__ __forceinline__ void mac2_vv(Acc& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
  ANEMOI_MAC2(ANEMOI_CV);
}
__device__ __forceinline__ void mac(Acc& acc, uint32_t a0, uint32_t b0) { ANEMOI_MAC1(ANEMOI_CV); }
__device__ __forceinline__ void mac4_vs(Acc& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2,
                                        uint32_t b2, uint32_t a3, uint32_t b3) {
  ANEMOI_MAC4(ANEMOI_CS);
}
__device__ __forceinline__ void mac2_vs(Acc& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
  ANEMOI_MAC2(ANEMOI_CS);
}
__device__ __forceinline__ void mac_k(Acc& acc, uint32_t a0, uint32_t b0) { ANEMOI_MAC1(ANEMOI_CS); }

__device__ __forceinline__ void acc_shift(Acc& acc) {
  acc.lo = (acc.lo >> 32) | ((uint64_t)acc.hi << 32);
  acc.hi = 0;
}
__device__ __forceinline__ uint32_t acc_lo(const Acc& acc) { return (uint32_t)acc.lo; }
// acc += 2 * x
__device__ __forceinline__ void acc_add2(Acc& acc, const Acc& x) {
  const uint64_t x2 = x.lo << 1;
  const uint32_t h2 = (x.hi << 1) | (uint32_t)(x.lo >> 63);
  const uint64_t s = acc.lo + x2;
  acc.hi += h2 + (s < x2 ? 1u : 0u);
  acc.lo = s;
}

template <int K, int J0, int CNT, int N>
__device__ __forceinline__ void mac_run_vv(Acc& acc, const uint32_t (&a)[N], const uint32_t (&b)[N]) {
  if constexpr (CNT >= 4) {
    mac4_vv(acc, a[J0], b[K - J0], a[J0 + 1], b[K - J0 - 1], a[J0 + 2], b[K - J0 - 2], a[J0 + 3], b[K - J0 - 3]);
    mac_run_vv<K, J0 + 4, CNT - 4, N>(acc, a, b);
  } else if constexpr (CNT >= 2) {
    mac2_vv(acc, a[J0], b[K - J0], a[J0 + 1], b[K - J0 - 1]);
    mac_run_vv<K, J0 + 2, CNT - 2, N>(acc, a, b);
  } else if constexpr (CNT == 1) {
    mac(acc, a[J0], b[K - J0]);
  }
}
template <class F, int K, int J0, int CNT, int N>
__device__ __forceinline__ void mac_run_vp(Acc& acc, const uint32_t (&m)[N]) {
  if constexpr (CNT >= 4) {
    mac4_vs(acc, m[J0], F::P[K - J0], m[J0 + 1], F::P[K - J0 - 1], m[J0 + 2], F::P[K - J0 - 2], m[J0 + 3],
            F::P[K - J0 - 3]);
    mac_run_vp<F, K, J0 + 4, CNT - 4, N>(acc, m);
  } else if constexpr (CNT >= 2) {
    mac2_vs(acc, m[J0], F::P[K - J0], m[J0 + 1], F::P[K - J0 - 1]);
    mac_run_vp<F, K, J0 + 2, CNT - 2, N>(acc, m);
  } else if constexpr (CNT == 1) {
    mac_k(acc, m[J0], F::P[K - J0]);
  }
}
#endif

// r = a - p if a >= p else a   (a < 2p)
template <class F>
__device__ __forceinline__ void fe_reduce_once(Fe<F::N>& a) {
  constexpr int N = F::N;
  uint32_t t[N];
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint64_t d = (uint64_t)a.l[i] - F::P[i] - borrow;
    t[i] = (uint32_t)d;
    borrow = (uint32_t)(d >> 63);
  }
#pragma unroll
  for (int i = 0; i < N; i++) a.l[i] = borrow ? a.l[i] : t[i];
}

// r = a + b mod p   (a, b < p)
template <class F>
__device__ __forceinline__ void fe_add(Fe<F::N>& r, const Fe<F::N>& a, const Fe<F::N>& b) {
  constexpr int N = F::N;
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint64_t s = (uint64_t)a.l[i] + b.l[i] + carry;
    r.l[i] = (uint32_t)s;
    carry = (uint32_t)(s >> 32);
  }
  // 2p < R for every supported field, so the sum has no carry out of limb N-1
  fe_reduce_once<F>(r);
}

// r = a + k mod p with k a compile-time/wave-uniform constant given as limbs
template <class F>
__device__ __forceinline__ void fe_add_const(Fe<F::N>& r, const Fe<F::N>& a, const uint32_t* __restrict__ k) {
  constexpr int N = F::N;
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint64_t s = (uint64_t)a.l[i] + k[i] + carry;
    r.l[i] = (uint32_t)s;
    carry = (uint32_t)(s >> 32);
  }
  fe_reduce_once<F>(r);
}

// r = a - b mod p   (a, b < p)
template <class F>
__device__ __forceinline__ void fe_sub(Fe<F::N>& r, const Fe<F::N>& a, const Fe<F::N>& b) {
  constexpr int N = F::N;
  uint32_t t[N], u[N];
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint64_t d = (uint64_t)a.l[i] - b.l[i] - borrow;
    t[i] = (uint32_t)d;
    borrow = (uint32_t)(d >> 63);
  }
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint64_t s = (uint64_t)t[i] + F::P[i] + carry;
    u[i] = (uint32_t)s;
    carry = (uint32_t)(s >> 32);
  }
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = borrow ? u[i] : t[i];
}

template <class F>
__device__ __forceinline__ void fe_dbl(Fe<F::N>& r, const Fe<F::N>& a) {
  fe_add<F>(r, a, a);
}

// One column of the product-scanning Montgomery product, columns unrolled by template recursion.
template <class F, bool SQR, int K>
struct Column {
  static constexpr int N = F::N;
  static constexpr int J0 = K < N ? 0 : K - N + 1;       // first index with both factors in range
  static constexpr int J1 = K < N ? K : N - 1;           // last such index
  __device__ static __forceinline__ void run(Acc& acc, const uint32_t (&a)[N], const uint32_t (&b)[N], uint32_t (&m)[N],
                                             uint32_t (&out)[N]) {
    if constexpr (!SQR) {
      mac_run_vv<K, J0, J1 - J0 + 1, N>(acc, a, b);
    } else {
      // off-diagonal products a[j]*a[K-j] for j < K-j, summed once and doubled
      constexpr int CROSS = (J1 - J0 + 1) / 2;
      if constexpr (CROSS > 0) {
        Acc x = {};
        mac_run_vv<K, J0, CROSS, N>(x, a, a);
        acc_add2(acc, x);
      }
      if constexpr ((K & 1) == 0) mac(acc, a[K / 2], a[K / 2]);
    }
    if constexpr (K < N) {
      mac_run_vp<F, K, 0, K, N>(acc, m);
      m[K] = acc_lo(acc) * F::kN0Inv;
      mac_k(acc, m[K], F::P[0]);
    } else {
      mac_run_vp<F, K, K - N + 1, 2 * N - 1 - K, N>(acc, m);
      out[K - N] = acc_lo(acc);
    }
    acc_shift(acc);
    if constexpr (K + 1 < 2 * N - 1) Column<F, SQR, K + 1>::run(acc, a, b, m, out);
  }
};

// Montgomery product r = a*b/R mod p.
//   FINAL = true : inputs < p (or < 2p), output canonical (< p)
//   FINAL = false: when F::kLazy (4p <= R) inputs < 2p -> output < 2p, no final subtraction
template <class F, bool FINAL>
__device__ __forceinline__ void mont_mul(Fe<F::N>& r, const Fe<F::N>& a, const Fe<F::N>& b) {
  constexpr int N = F::N;
  uint32_t m[N], out[N];
  Acc acc = {};
  Column<F, false, 0>::run(acc, a.l, b.l, m, out);
  out[N - 1] = acc_lo(acc);
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = out[i];
  if (FINAL || !F::kLazy) fe_reduce_once<F>(r);
}

// Montgomery square r = a*a/R mod p; same contract as mont_mul.
template <class F, bool FINAL>
__device__ __forceinline__ void mont_sqr(Fe<F::N>& r, const Fe<F::N>& a) {
  constexpr int N = F::N;
  uint32_t m[N], out[N];
  Acc acc = {};
  Column<F, true, 0>::run(acc, a.l, a.l, m, out);
  out[N - 1] = acc_lo(acc);
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = out[i];
  if (FINAL || !F::kLazy) fe_reduce_once<F>(r);
}

// g * x for the S-box / MDS generator (reference: mul_by_generator, src/traits.rs:78-91)
template <class F>
__device__ __forceinline__ void fe_mul_g(Fe<F::N>& r, const Fe<F::N>& x) {
  Fe<F::N> d;
  if (F::kG == 2) {
    fe_dbl<F>(r, x);
  } else if (F::kG == 3) {
    fe_dbl<F>(d, x);
    fe_add<F>(r, d, x);
  } else if (F::kG == 5) {
    fe_dbl<F>(d, x);
    fe_dbl<F>(d, d);
    fe_add<F>(r, d, x);
  } else if (F::kG == 7) {
    fe_dbl<F>(d, x);
    fe_add<F>(d, d, x);
    fe_dbl<F>(d, d);
    fe_add<F>(r, d, x);
  } else if (F::kG == 15) {
    fe_dbl<F>(d, x);
    fe_dbl<F>(d, d);
    fe_dbl<F>(d, d);
    fe_dbl<F>(d, d);
    fe_sub<F>(r, d, x);
  } else {
    Fe<F::N> g;
#pragma unroll
    for (int i = 0; i < F::N; i++) g.l[i] = F::GMont[i];
    mont_mul<F, true>(r, x, g);
  }
}

// Policy adapter used by anemoi_perm.h / anemoi_kernels.h (same interface as Arith29 in mont29.h).
// Values held between operations are canonical (< p); only the exponentiation chain is lazy.
template <class F>
struct Arith32 {
  static constexpr int NL = F::N;
  static constexpr int NABI = F::N;
  static constexpr int NQ = F::N / 4;
  static constexpr bool kLoose = false;
  using Fe = anemoi::Fe<F::N>;

  __device__ static __forceinline__ void mul(Fe& r, const Fe& a, const Fe& b) { mont_mul<F, true>(r, a, b); }
  __device__ static __forceinline__ void sqr(Fe& r, const Fe& a) { mont_sqr<F, true>(r, a); }
  __device__ static __forceinline__ void emul(Fe& r, const Fe& a, const Fe& b) { mont_mul<F, false>(r, a, b); }
  __device__ static __forceinline__ void esqr(Fe& r, const Fe& a) { mont_sqr<F, false>(r, a); }
  __device__ static __forceinline__ void efinish(Fe& a) {
    if (F::kLazy) fe_reduce_once<F>(a);
  }
  __device__ static __forceinline__ void add(Fe& r, const Fe& a, const Fe& b) { fe_add<F>(r, a, b); }
  __device__ static __forceinline__ void add_k(Fe& r, const Fe& a, const uint32_t* __restrict__ k) {
    fe_add_const<F>(r, a, k);
  }
  __device__ static __forceinline__ void sub(Fe& r, const Fe& a, const Fe& b) { fe_sub<F>(r, a, b); }
  __device__ static __forceinline__ void mul_g(Fe& r, const Fe& x) { fe_mul_g<F>(r, x); }
  __device__ static __forceinline__ void settle(Fe&) {}
  __device__ static __forceinline__ void add_delta(Fe& r, const Fe& a) { fe_add_const<F>(r, a, F::Delta); }
  __device__ static __forceinline__ void set_one(Fe& x) {
#pragma unroll
    for (int i = 0; i < F::N; i++) x.l[i] = F::One[i];
  }
  __device__ static __forceinline__ void set_zero(Fe& x) {
#pragma unroll
    for (int i = 0; i < F::N; i++) x.l[i] = 0;
  }
  __device__ static __forceinline__ void from_abi(Fe& r, const uint32_t (&w)[NABI]) {
#pragma unroll
    for (int i = 0; i < F::N; i++) r.l[i] = w[i];
  }
  __device__ static __forceinline__ void to_abi(uint32_t (&w)[NABI], const Fe& a) {
#pragma unroll
    for (int i = 0; i < F::N; i++) w[i] = a.l[i];
  }
  __device__ static __forceinline__ void from_int(Fe& r, const uint32_t (&w)[NABI]) {
    Fe v, r2;
#pragma unroll
    for (int i = 0; i < F::N; i++) {
      v.l[i] = w[i];
      r2.l[i] = F::R2[i];
    }
    mont_mul<F, true>(r, v, r2);
  }
  __device__ static __forceinline__ void lds_store(uint4* base, int stride, const Fe& v) {
#pragma unroll
    for (int q = 0; q < NQ; q++)
      base[q * stride] = make_uint4(v.l[4 * q], v.l[4 * q + 1], v.l[4 * q + 2], v.l[4 * q + 3]);
  }
  __device__ static __forceinline__ void lds_load(const uint4* base, int stride, Fe& v) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      uint4 t = base[q * stride];
      v.l[4 * q] = t.x;
      v.l[4 * q + 1] = t.y;
      v.l[4 * q + 2] = t.z;
      v.l[4 * q + 3] = t.w;
    }
  }
  static const uint32_t* host_ark(int width, bool d) {
    return width == 2 ? (d ? F::ArkD21 : F::ArkC21) : (d ? F::ArkD43 : F::ArkC43);
  }
};

}  // namespace anemoi
