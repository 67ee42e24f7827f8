// anemoi_perm.h -- the Anemoi round function and permutation, one state per lane.
//
// Follows the reference's generic permutation (src/traits.rs):
//   ark_layer   :111-125   state[i] += C[r*c+i]; state[c+i] += D[r*c+i]
//   mds_layer   :136-157   arms NUM_COLUMNS = 1 and 2 (the only shipped ones)
//   sbox_layer  :326-358   x -= g*y^2 ; y -= x^(1/alpha) ; x += g*y^2 + delta
//   round       :361-367   ark ; mds ; sbox
//   permutation :370-378   NUM_ROUNDS rounds, then one more mds_layer
// and replaces the per-field hard-coded addition chain `exp_by_inv_alpha` (src/<field>/sbox.rs) by
// a sliding-window exponentiation whose table of odd powers lives in LDS (one private column per
// lane, 16-byte interleaved so every ds_read_b128/ds_write_b128 is conflict-free).  x^INV_ALPHA is a
// canonical field value, so any exponentiation schedule is bit-identical to the reference's chain.
#pragma once
#include "field_consts_gen.h"
#include "mont32.h"

namespace anemoi {

// Per-lane window table in LDS.  Entry e, 16-byte quad q of this lane sits at
// base[(e * Q + q) * stride]; `base` already points at the lane's own slot.
template <int N>
struct LdsTable {
  static constexpr int Q = N / 4;
  uint4* base;
  int stride;  // in uint4 = number of lanes sharing the table block

  __device__ __forceinline__ void store(int e, const Fe<N>& v) const {
#pragma unroll
    for (int q = 0; q < Q; q++)
      base[(e * Q + q) * stride] = make_uint4(v.l[4 * q], v.l[4 * q + 1], v.l[4 * q + 2], v.l[4 * q + 3]);
  }
  __device__ __forceinline__ void load(int e, Fe<N>& v) const {
#pragma unroll
    for (int q = 0; q < Q; q++) {
      uint4 t = base[(e * Q + q) * stride];
      v.l[4 * q] = t.x;
      v.l[4 * q + 1] = t.y;
      v.l[4 * q + 2] = t.z;
      v.l[4 * q + 3] = t.w;
    }
  }
};

// Per-(field, width) constants in device memory, uploaded once by the host context:
//   ark   : C then D, each cols*rounds elements of N Montgomery limbs (uniform -> scalar loads)
//   sched : sliding-window schedule, pairs (squarings, table index | 255), `steps` of them
struct PermConsts {
  const uint32_t* ark_c;
  const uint32_t* ark_d;
  const uint8_t* sched;
  int steps;
  int first;
};

// r = x^INV_ALPHA (canonical in, canonical out).  WIN-bit sliding window over odd powers.
template <class F, int WIN>
__device__ __forceinline__ void exp_inv_alpha(Fe<F::N>& r, const Fe<F::N>& x, const PermConsts& pc,
                                              const LdsTable<F::N>& tab) {
  constexpr int E = 1 << (WIN - 1);
  Fe<F::N> x2, t;
  mont_sqr<F, false>(x2, x);
  t = x;
  tab.store(0, t);
#pragma nounroll
  for (int i = 1; i < E; i++) {
    mont_mul<F, false>(t, t, x2);
    tab.store(i, t);
  }
  Fe<F::N> acc;
  tab.load(pc.first, acc);
#pragma nounroll
  for (int s = 0; s < pc.steps; s++) {
    const int nsq = pc.sched[2 * s], idx = pc.sched[2 * s + 1];
#pragma nounroll
    for (int q = 0; q < nsq; q++) mont_sqr<F, false>(acc, acc);
    if (idx != 255) {
      tab.load(idx, t);
      mont_mul<F, false>(acc, acc, t);
    }
  }
  if (F::kLazy) fe_reduce_once<F>(acc);
  r = acc;
}

// Flystel S-box on one column (src/traits.rs:326-358)
template <class F, int WIN>
__device__ __forceinline__ void flystel(Fe<F::N>& x, Fe<F::N>& y, const PermConsts& pc, const LdsTable<F::N>& tab) {
  Fe<F::N> t, u;
  mont_sqr<F, true>(t, y);
  fe_mul_g<F>(u, t);
  fe_sub<F>(x, x, u);
  exp_inv_alpha<F, WIN>(t, x, pc, tab);
  fe_sub<F>(y, y, t);
  mont_sqr<F, true>(t, y);
  fe_mul_g<F>(u, t);
  fe_add<F>(x, x, u);
  fe_add_const<F>(x, x, F::Delta);
}

// Linear layer (src/traits.rs:136-157)
template <class F, int W>
__device__ __forceinline__ void mds_layer(Fe<F::N> (&st)[W]) {
  if (W == 2) {
    fe_add<F>(st[1], st[1], st[0]);
    fe_add<F>(st[0], st[0], st[1]);
  } else {
    Fe<F::N> t;
    fe_mul_g<F>(t, st[1]);
    fe_add<F>(st[0], st[0], t);
    fe_mul_g<F>(t, st[0]);
    fe_add<F>(st[1], st[1], t);
    fe_mul_g<F>(t, st[2]);
    fe_add<F>(st[3], st[3], t);
    fe_mul_g<F>(t, st[3]);
    fe_add<F>(st[2], st[2], t);
    t = st[2];
    st[2] = st[3];
    st[3] = t;
    fe_add<F>(st[2], st[2], st[0]);
    fe_add<F>(st[3], st[3], st[1]);
    fe_add<F>(st[0], st[0], st[2]);
    fe_add<F>(st[1], st[1], st[3]);
  }
}

template <class F>
__device__ __forceinline__ void fe_add_global(Fe<F::N>& r, const uint32_t* __restrict__ k) {
  uint32_t c[F::N];
#pragma unroll
  for (int i = 0; i < F::N; i++) c[i] = k[i];
  fe_add_const<F>(r, r, c);
}

// Full permutation (src/traits.rs:370-378)
template <class F, int W, int WIN>
__device__ __forceinline__ void permutation(Fe<F::N> (&st)[W], const PermConsts& pc, const LdsTable<F::N>& tab) {
  constexpr int C = W / 2;
  constexpr int R = W == 2 ? F::kRounds21 : F::kRounds43;
#pragma nounroll
  for (int r = 0; r < R; r++) {
#pragma unroll
    for (int i = 0; i < C; i++) {
      fe_add_global<F>(st[i], pc.ark_c + (r * C + i) * F::N);
      fe_add_global<F>(st[C + i], pc.ark_d + (r * C + i) * F::N);
    }
    mds_layer<F, W>(st);
    // columns spelled out: the S-box body is too large for `#pragma unroll`, and a rolled loop
    // would index the register-resident state dynamically (scratch)
    flystel<F, WIN>(st[0], st[C], pc, tab);
    if (C == 2) flystel<F, WIN>(st[C - 1], st[W - 1], pc, tab);
  }
  mds_layer<F, W>(st);
}

}  // namespace anemoi
