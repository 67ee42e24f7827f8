// anemoi_perm.h -- the Anemoi round function and permutation, one state per lane.
//
// Follows the reference's generic permutation (src/traits.rs):
//   ark_layer   :111-125   state[i] += C[r*c+i]; state[c+i] += D[r*c+i]
//   mds_layer   :136-157   arms NUM_COLUMNS = 1 and 2 (the only shipped ones)
//   sbox_layer  :326-358   x -= g*y^2 ; y -= x^(1/alpha) ; x += g*y^2 + delta
//   round       :361-367   ark ; mds ; sbox
//   permutation :370-378   NUM_ROUNDS rounds, then one more mds_layer
// and replaces the per-field hard-coded addition chain `exp_by_inv_alpha` (src/<field>/sbox.rs) by
// a sliding-window exponentiation: x^INV_ALPHA is a canonical field value, so any exponentiation
// schedule is bit-identical to the reference's chain.  x itself stays in registers (the S-box
// needs it afterwards anyway); the higher odd powers x^3, x^5, .. live in LDS, one private column
// per lane, 16-byte interleaved so every ds_read_b128 / ds_write_b128 is conflict-free.
//
// The arithmetic is a policy `A` (Arith32 in mont32.h, Arith29 in mont29.h).  With a "loose"
// policy (Arith29) additions do not reduce; values are bounded by the bookkeeping in the comments
// below (units of p), and `settle` brings a value back below 2p.
#pragma once
#include <type_traits>

#include "field_consts_gen.h"
#include "mont29.h"
#include "mont32.h"

// ANEMOI_XDIGITS_ON (build_config.h): extra window digits in VGPRs (exp_inv_alpha); 0 = plain window (A/B builds)

namespace anemoi {

// Compile-time loop over state elements: bodies that inline a whole Montgomery product are too big
// for `#pragma unroll`, and a rolled loop would index the register-resident state dynamically
// (scratch memory).
template <int I, int N, class Fn>
__device__ __forceinline__ void static_for(Fn&& fn) {
  if constexpr (I < N) {
    fn(std::integral_constant<int, I>{});
    static_for<I + 1, N>(fn);
  }
}

// Every field runs on unsaturated limbs (mont29.h: 13 limbs of 30 bits for 381/377 bits, 9 of 29 for 253..255 bits).
// A/B builds only: ANEMOI_ARITH32_FIELDS (a bit mask of field ids) puts selected fields on the saturated 32-bit-limb
// arithmetic of round 1 (mont32.h; 4.3 M against 10.6 M BLS12-381 compressions/s).
#if defined(ANEMOI_BOUNDS_WALK)
// tests/cpp/bounds_walk: the kernel bodies compiled for the HOST over an arithmetic that carries a bound of every value,
// so that the lazy-reduction bounds are walked on the code the kernels are made of (tests/test_bounds_walk.py)
template <class F>
struct BoundsWalkArith;
template <int FIELD>
using ArithFor = BoundsWalkArith<FieldC<FIELD>>;
#elif ANEMOI_AB_BUILD
template <int FIELD>
using ArithFor = std::conditional_t<((ANEMOI_ARITH32_FIELDS >> FIELD) & 1) == 0, Arith29<FieldC<FIELD>>,
                                    Arith32<FieldC<FIELD>>>;
#else
template <int FIELD>
using ArithFor = Arith29<FieldC<FIELD>>;
#endif

// Per-lane window table in LDS: entries 1.. = x^3, x^5, ...; entry e, slot q of this lane at
// base[((e-1) * NQ + q) * stride].
template <class A>
struct LdsTable {
  uint4* base;  // already offset to the lane's own column
  int stride;   // uint4 between consecutive slots = lanes per block
  __device__ __forceinline__ void store(int e, const typename A::Fe& v) const {
    A::lds_store(base + (e - 1) * A::NQ * stride, stride, v);
  }
  __device__ __forceinline__ void load(int e, typename A::Fe& v) const {
    A::lds_load(base + (e - 1) * A::NQ * stride, stride, v);
  }
};

// Per-(field, width) constants in device memory, uploaded once by the host context:
//   ark   : C then D, each cols*rounds elements of A::NL limbs in A's Montgomery form (wave-uniform
//           addresses -> scalar loads)
//   sched : sliding-window schedule, `steps` words (squarings | op << 8); op = table index, or 255
//           (no multiplication), 253 (tmp = acc), 254 (multiply by tmp) -- tools/gen_params.py.  One
//           32-bit word per step so that the wave-uniform fetch is a scalar load (s_load_dword):
//           byte entries made hipcc fetch them with a vector load and wait for it on every step
struct PermConsts {
  const uint32_t* ark_c;
  const uint32_t* ark_d;
  const uint32_t* sched;
  int steps;
  int first;
  // the plain WIN-bit window schedule, for kernels that cannot spare the registers of the extra digits
  // (k_merkle_climb): equal to sched / steps / first when the field has no extra digits
  const uint32_t* sched_plain;
  int steps_plain;
  int first_plain;
  // schedule for the wave-cooperative kernels (window F::kCoopWin: their table costs one LDS word per entry)
  const uint32_t* sched5;
  int steps5;
  int first5;
  // the instance's round constants in the cooperative kernels' limb layout (F::Coop)
  const uint32_t* coop_c;
  const uint32_t* coop_d;
  // ... and in the two-row fold layout (F::Fold, coop2d.h); both widths
  const uint32_t* fold_c;
  const uint32_t* fold_d;
  // host-side routing only: SIMDs of the device these tables live on (4 per CU); the automatic cut-offs of the
  // latency kernels are multiples of it (anemoi_coop_kernels.h)
  int simds;
};

// Source `SRC` of the extra digits' build programme (tools/gen_params.py): 0 = x, 1 = x^2, 2..4 = the LDS table
// entries x^3, x^5, x^7, 5 = the first extra digit.
template <class A, int SRC>
__device__ __forceinline__ void xdigit_src(typename A::Fe& dst, const typename A::Fe& x, const typename A::Fe& x2,
                                           const typename A::Fe& e0, const LdsTable<A>& tab) {
  if constexpr (SRC == 0) dst = x;
  else if constexpr (SRC == 1) dst = x2;
  else if constexpr (SRC <= 4) tab.load(SRC - 1, dst);
  else dst = e0;
}

// x^d for extra digit XI, by its build programme: LOAD a ; [MUL b] ; [SQR k] ; [MUL r]
template <class F, class A, int XI>
__device__ __forceinline__ void xdigit_build(typename A::Fe& r, const typename A::Fe& x, const typename A::Fe& x2,
                                             const typename A::Fe& e0, const LdsTable<A>& tab) {
  constexpr int OFF = XI == 0 ? 0 : F::kXProgLen[0];
  static_for<0, F::kXProgLen[XI]>([&](auto k) {
    constexpr int op = F::kXProgOp[OFF + k], arg = F::kXProgArg[OFF + k];
    if constexpr (op == 0) {
      xdigit_src<A, arg>(r, x, x2, e0, tab);
    } else if constexpr (op == 1) {
#pragma nounroll
      for (int q = 0; q < arg; q++) A::esqr(r, r);
    } else {
      typename A::Fe s;
      xdigit_src<A, arg>(s, x, x2, e0, tab);
      A::emul(r, r, s);
    }
  });
}

// r = x^INV_ALPHA.  WIN-bit sliding window over odd powers; table entry 0 is x itself (registers).
// In: x < 2^12 p (loose) or canonical.  Out: < 2p (loose) or canonical.
// With WIN = 3 and F::kXDigits > 0 the table {x, x^3, x^5, x^7} is extended by one or two more powers x^d held
// in VGPRs (digits and schedule chosen per field by tools/gen_params.py: e.g. 51 = 0b110011 and 59 for BLS12-381,
// 468 products instead of 478 -- the LDS budget of 3-4 waves per SIMD stops at three entries, the register
// budget has room for these); a multiplication by such a digit takes its operand straight from the registers.
// One word of a wave-uniform device table (the exponent schedule) through the SCALAR cache.  Written as a plain
// `table[i]`, hipcc cannot prove that nothing in the kernel clobbers the table and fetches it with a vector load +
// v_readfirstlane + s_waitcnt vmcnt(0): a vector-memory round trip in front of every step of the exponentiation
// (54-80 per S-box).  Three or four resident wavefronts hide that; a lone wavefront (the latency kernels, the middle
// levels of a Merkle tree) or two (config 3) do not: in-process A/B (profiles/r03/ab_scalar_schedule_loads.txt) one
// cooperative compression -1.5 %, 4 096 of them -5 %, lane-private batches of 2^15 .. 2^16 items -2.5 .. -8.8 %, full
// batches unchanged.  The tables are written once by the host before any launch, so reading them through the constant
// address space is exact.  (Restructuring the exponentiation into ONE loop with ONE multiplication statement, which
// removes the ~39 v_mov per step hipcc places around the squaring loop and the per-operand multiplications, was built
// and measured in the same A/B: no gain on full batches -- v_mov is nearly free next to v_mad_u64_u32 -- and a loss
// on underfilled ones, because the schedule load then sits right in front of its use.  Not adopted.)
__device__ __forceinline__ uint32_t uniform_word(const uint32_t* table, int i) {
  typedef const uint32_t __attribute__((address_space(4))) * ConstWords;
  return ((ConstWords)(uintptr_t)table)[i];
}

template <class F, class A, int WIN, bool USEX = true>
__device__ __forceinline__ void exp_inv_alpha(typename A::Fe& r, const typename A::Fe& x, const PermConsts& pc,
                                              const LdsTable<A>& tab) {
  constexpr int E = 1 << (WIN - 1);
  constexpr bool kX = USEX && WIN == 3 && F::kXDigits > 0 && ANEMOI_XDIGITS_ON;
  const uint32_t* const sched = kX ? pc.sched : pc.sched_plain;
  const int steps = kX ? pc.steps : pc.steps_plain, first = kX ? pc.first : pc.first_plain;
  typename A::Fe x2, t, acc;
  A::esqr(x2, x);
  t = x;
#pragma nounroll
  for (int i = 1; i < E; i++) {
    A::emul(t, t, x2);
    tab.store(i, t);
  }
  [[maybe_unused]] typename A::Fe e0, e1;  // the extra digits
  if constexpr (kX) {
    xdigit_build<F, A, 0>(e0, x, x2, e0, tab);
    if constexpr (F::kXDigits > 1) xdigit_build<F, A, 1>(e1, x, x2, e0, tab);
  }
  if (first == 0) acc = x;
  else if (!kX || first < E) tab.load(first, acc);
  else if (F::kXDigits > 1 && first == E + 1) acc = e1;
  else acc = e0;
  [[maybe_unused]] typename A::Fe tmp;  // leading-run doubling (Pallas / Vesta), see sliding_window()
  if constexpr (F::kChainTmp) tmp = acc;
#pragma nounroll
  for (int s = 0; s < steps; s++) {
    const uint32_t word = uniform_word(sched, s);
    const int nsq = word & 0xff, idx = word >> 8;
    if constexpr (F::kChainTmp) {
      if (idx == 253) {
        tmp = acc;
        continue;
      }
    }
#pragma nounroll
    for (int q = 0; q < nsq; q++) A::esqr(acc, acc);
    if (F::kChainTmp && idx == 254) {
      A::emul(acc, acc, tmp);
    } else if (kX && idx == E) {
      A::emul(acc, acc, e0);
    } else if (kX && F::kXDigits > 1 && idx == E + 1) {
      A::emul(acc, acc, e1);
    } else if (idx != 255) {
      if (idx == 0) t = x;
      else tab.load(idx, t);
      A::emul(acc, acc, t);
    }
  }
  A::efinish(acc);
  r = acc;
}

// Flystel S-box on one column (src/traits.rs:326-358).
// Loose bounds: nothing here reduces.  A Montgomery product of inputs < A p and < B p is < (A B / H + 1) p with
// H = R'/p, a subtraction adds the pad (kSubK p: 8 p for BLS12-381 on 30-bit limbs, 64 p on 29-bit limbs of the
// 381/377-bit fields, 4 p for the "tight" layouts whose g * x is a product), and the caller settles x and y once per
// round after the linear layer.  What every statement below meets -- largest operands, largest result, share of its
// limit, for every field and every kernel that inlines it -- is in BOUNDS.md next to this file: generated by walking
// THIS code over a bound-carrying arithmetic (tests/cpp/bounds_walk, tools/bounds_walk.py) and checked by
// tests/test_bounds_walk.py, which fails when a pad, a limb count, a schedule or the order of two statements here
// breaks a precondition.  Orders of magnitude (lane-private layouts): the S-box input x' is < 9.1 p on BLS12-381
// (limit of the 30-bit assembly: 16 p), < 5.3 p on the tight fields (A B = 27 of H = 70.7 on Jubjub); every
// subtrahend is a product or g times one (<= 2.0x p, pads 4 p / 8 p).
template <class F, class A, int WIN, bool USEX = true>
__device__ __forceinline__ void flystel(typename A::Fe& x, typename A::Fe& y, const PermConsts& pc,
                                        const LdsTable<A>& tab) {
  typename A::Fe t, u;
  A::sqr(t, y);
  A::mul_g(u, t);
  A::sub(x, x, u);
  exp_inv_alpha<F, A, WIN, USEX>(t, x, pc, tab);
  A::sub(y, y, t);
  A::sqr(t, y);
  A::mul_g(u, t);
  A::add(x, x, u);
  A::add_delta(x, x);
}

// Linear layer (src/traits.rs:136-157): additions only (g * x by limb scaling or as a product, mont29.h), then ONE
// settle per state element -- valid while the sums stay below H p, and below R' as limb vectors.  Largest sums
// (BOUNDS.md, walked): W = 2: < 21 p on Jubjub (H = 70.7), < 35 p on BLS12-381 (H = 630); W = 4: < 24.5 p on the
// 9-limb fields, < 240 p on BLS12-381 (g = 2 by doubling: 85.5 p + 153.2 p in the last PHT addition) -- 38 % of R' at most.
template <class F, class A, int W>
__device__ __forceinline__ void mds_layer(typename A::Fe (&st)[W]) {
  if (W == 2) {
    A::add(st[1], st[1], st[0]);
    A::add(st[0], st[0], st[1]);
  } else {
    typename A::Fe t;
    A::mul_g(t, st[1]);
    A::add(st[0], st[0], t);
    A::mul_g(t, st[0]);
    A::add(st[1], st[1], t);
    A::mul_g(t, st[2]);
    A::add(st[3], st[3], t);
    A::mul_g(t, st[3]);
    A::add(st[2], st[2], t);
    t = st[2];
    st[2] = st[3];
    st[3] = t;
    A::add(st[2], st[2], st[0]);
    A::add(st[3], st[3], st[1]);
    A::add(st[0], st[0], st[2]);
    A::add(st[1], st[1], st[3]);
  }
  if (A::kLoose) static_for<0, W>([&](auto i) { A::settle(st[i]); });
}

template <class A>
__device__ __forceinline__ void add_global(typename A::Fe& r, const uint32_t* __restrict__ k) {
  uint32_t c[A::NL];
#pragma unroll
  for (int i = 0; i < A::NL; i++) c[i] = k[i];
  A::add_k(r, r, c);
}

template <class F, class A, int W, int WIN, bool USEX = true>
__device__ __forceinline__ void sbox_layer(typename A::Fe (&st)[W], const PermConsts& pc, const LdsTable<A>& tab) {
  // columns spelled out: the S-box body is too large to unroll by pragma, and a rolled loop would
  // index the register-resident state dynamically (scratch)
  flystel<F, A, WIN, USEX>(st[0], st[W / 2], pc, tab);
  if (W == 4) flystel<F, A, WIN, USEX>(st[1], st[3], pc, tab);
}

// Two wavefronts that share a SIMD do not share it fairly: the arbiter prefers one of them (it runs at the lone-wave
// rate, one instruction per ~6.5 cycles, the other gets the remaining slots), so in a launch with two resident
// wavefronts per SIMD doing equal work -- config 3: 2 048 wavefronts on 1 024 SIMDs -- the preferred one finishes after
// 0.73 of the time and its partner then runs ALONE at 61 % of the SIMD's issue rate: SQ_WAVE_CYCLES shows an average
// residency of 0.79 and the launch is ~12 % slower than two fair halves would be (profiles/r03/pmc_configs.json).
// Alternating the wave priority round by round, in opposite phase for even and odd wave slots of a SIMD, makes the
// two take turns and finish together.  With 3-4 resident wavefronts (the full batches) it changes nothing.
__device__ __forceinline__ uint32_t wave_slot() {
#if ANEMOI_ALT_PRIO
  return __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 1u;   // HW_ID.wave_id[3:0]: the slot on its SIMD
#else
  return 0;
#endif
}
__device__ __forceinline__ void alternate_priority(int round, uint32_t slot) {
#if ANEMOI_ALT_PRIO
  if ((uint32_t(round) + slot) & 1u) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
#endif
}

// Full permutation (src/traits.rs:370-378)
template <class F, class A, int W, int WIN, bool USEX = true>
__device__ __forceinline__ void permutation(typename A::Fe (&st)[W], const PermConsts& pc, const LdsTable<A>& tab) {
  constexpr int C = W / 2;
  constexpr int R = W == 2 ? F::kRounds21 : F::kRounds43;
  const uint32_t slot = wave_slot();
#pragma nounroll
  for (int r = 0; r < R; r++) {
    alternate_priority(r, slot);
    static_for<0, C>([&](auto i) {
      add_global<A>(st[i], pc.ark_c + (r * C + i) * A::NL);
      add_global<A>(st[C + i], pc.ark_d + (r * C + i) * A::NL);
    });
    mds_layer<F, A, W>(st);
    sbox_layer<F, A, W, WIN, USEX>(st, pc, tab);
  }
  mds_layer<F, A, W>(st);
}

// ---- Anemoi-4-3 with the two columns of a state on two adjacent lanes ------------------------------
// Lane 2i holds (x0, y0) = (state[0], state[2]) of state i, lane 2i+1 holds (x1, y1) = (state[1],
// state[3]).  The two S-boxes of a round (99 % of the work) are independent, so a state advances on
// two lanes at once: half the latency per permutation and twice the wavefronts for a given batch,
// which is what small batches need (one wavefront alone issues only one VALU instruction per ~6
// cycles).  Only the linear layer couples the columns; it exchanges values with the neighbour lane
// through DPP quad_perm:[1,0,3,2].
template <class A>
__device__ __forceinline__ void fe_exchange(typename A::Fe& out, const typename A::Fe& in) {
#pragma unroll
  for (int i = 0; i < A::NL; i++)
    out.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)in.l[i], 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false);
}

template <class A>
__device__ __forceinline__ void fe_select(typename A::Fe& r, bool take_a, const typename A::Fe& a,
                                          const typename A::Fe& b) {
#pragma unroll
  for (int i = 0; i < A::NL; i++) r.l[i] = take_a ? a.l[i] : b.l[i];
}

// mds_layer arm NUM_COLUMNS = 2 (src/traits.rs:143-157), statement by statement:
//   s0 += g s1 ; s1 += g s0 ; s3 += g s2 ; s2 += g s3 ; swap(s2, s3) ; s2 += s0 ; s3 += s1 ; s0 += s2 ; s1 += s3
template <class F, class A>
__device__ __forceinline__ void mds_pair(typename A::Fe& x, typename A::Fe& y, bool odd) {
  // The reference's four generator statements are two independent chains, (s0 += g s1 ; s1 += g s0) on the x
  // row and (s3 += g s2 ; s2 += g s3) on the y row.  The first statement of each chain belongs to a different
  // lane of the pair (s0: even lane's x, s3: odd lane's y), and so does the second (s1: odd lane's x, s2: even
  // lane's y) -- so each lane runs ONE g-multiplication per step on its own operand, two per layer instead
  // of four (for the fields whose g * x is a full Montgomery product that is 2 of ~320 products per round).
  typename A::Fe px, py, p, t, s;
  fe_exchange<A>(px, x);
  fe_exchange<A>(py, y);
  fe_select<A>(p, odd, py, px);  // even: s1 (the odd lane's x); odd: s2 (the even lane's y)
  A::mul_g(t, p);
  A::add(s, x, t);
  fe_select<A>(x, !odd, s, x);   // even: s0 += g s1
  A::add(s, y, t);
  fe_select<A>(y, odd, s, y);    // odd:  s3 += g s2
  fe_exchange<A>(px, x);
  fe_exchange<A>(py, y);
  fe_select<A>(p, odd, px, py);  // odd: the updated s0 (even lane's x); even: the updated s3 (odd lane's y)
  A::mul_g(t, p);
  A::add(s, x, t);
  fe_select<A>(x, odd, s, x);    // odd:  s1 += g s0
  A::add(s, y, t);
  fe_select<A>(y, !odd, s, y);   // even: s2 += g s3
  fe_exchange<A>(p, y);  // swap(s2, s3)
  y = p;
  A::add(y, y, x);  // s2 += s0 ; s3 += s1
  A::add(x, x, y);  // s0 += s2 ; s1 += s3
  if (A::kLoose) {
    A::settle(x);
    A::settle(y);
  }
}

template <class F, class A, int WIN>
__device__ __forceinline__ void permutation_pair(typename A::Fe& x, typename A::Fe& y, bool odd, const PermConsts& pc,
                                                 const LdsTable<A>& tab) {
  constexpr int R = F::kRounds43;
  const int col = odd ? 1 : 0;
  const uint32_t slot = wave_slot();
#pragma nounroll
  for (int r = 0; r < R; r++) {
    alternate_priority(r, slot);
    add_global<A>(x, pc.ark_c + (r * 2 + col) * A::NL);
    add_global<A>(y, pc.ark_d + (r * 2 + col) * A::NL);
    mds_pair<F, A>(x, y, odd);
    flystel<F, A, WIN>(x, y, pc, tab);
  }
  mds_pair<F, A>(x, y, odd);
}

}  // namespace anemoi
