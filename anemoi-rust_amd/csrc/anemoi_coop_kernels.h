// anemoi_coop_kernels.h -- the LATENCY kernels: small batches on the cooperative arithmetics -- the two-row fold product of
// coop2d.h (an element on a pair of 16-lane DPP rows; LPR = 32) and the digit-serial scan of coop29.h (one 29-bit limb per
// lane, an element per row; LPR = 16) -- and the cut-offs that route a launch to them.
//
//   k_jive2_coop<F, LPR>       Jive::compress 2-1 / Sponge::merge   src/<f>/anemoi_2_1/hasher.rs:86-103    2 / 4 items per wavefront
//   k_jive2_coop<F, 64>        the same, one item per wavefront (four-row fold / rounds 1-2's scan; A/B and parity only)
//   k_jive4_coop<F, K, LPR>    Jive::compress(_k) 4-3               src/<f>/anemoi_4_3/hasher.rs:148-179   1 / 2 states per wavefront
//   k_permutation_coop         Anemoi::permutation                  src/traits.rs:370-378
//   k_merkle_climb_coop        authentication-path verification     (depth x merge)
//   k_sponge_coop              Sponge::hash / hash_field            anemoi_2_1/hasher.rs:18-85, anemoi_4_3/hasher.rs:19-129
//
// All of them are built on coop_permutation (one column (x, y) per element; the 4-3 linear layer couples two columns) and
// coop_flystel (the S-box with its sliding-window exponentiation, table in LDS).  Same round function and the same
// bound bookkeeping as the lane-private kernels of anemoi_kernels.h; every kernel is forced for every size and field
// against the oracle by tests/test_gpu_parity.py.
#pragma once
// included from anemoi_kernels.h (after kBlock, PermConsts and uniform_word)

namespace anemoi {

// Lanes per item -> the cooperative arithmetic:
//   16   the digit-serial scan of coop29.h, one item per DPP row (four per wavefront)
//   32   the two-row fold product of coop2d.h (two items per wavefront)
//   64   one item per wavefront (`make AB=1` libraries only): the FOUR-row fold product where it exists (11-limb fields), else
//        the one-element scan of coop29.h (rounds 1-2's kernel)
#ifdef ANEMOI_BOUNDS_WALK   // host walk of the bounds (tests/cpp/bounds_walk): a recording arithmetic in place of both
template <class F, int LPI>
struct BoundsWalkCoop;
template <class F, int LPI, bool FOUR_ROWS = false>
struct CoopArith { using type = BoundsWalkCoop<F, LPI>; };
#else
template <class F, int LPI, bool FOUR_ROWS = (F::Fold::Q4 > 0)>
struct CoopArith { using type = Coop29<F, LPI>; };
template <class F, bool FR>
struct CoopArith<F, 32, FR> { using type = Coop2d<F, 2>; };
#if ANEMOI_AB_BUILD
template <class F>
struct CoopArith<F, 64, true> { using type = Coop2d<F, 4>; };
#endif
#endif

// ---- wave-cooperative Jive 2-to-1 compression (coop29.h, coop2d.h) ------------------------------------
// Latency path: small batches (the top levels of a Merkle tree, a single Jive::compress / Sponge::merge call).
// Same round function and the same bound bookkeeping as the lane-private kernels; the window table (F::kCoopWin
// bits, one word per lane and entry) sits in LDS.  Two layouts of the same arithmetic:
//   LPR = 16  FOUR items per wavefront, one per 16-lane DPP row; everything row-local (a_i and the quotient digit
//             by DPP row broadcast, the digit computed on the VALU).  The shipped latency kernel: one BLS12-381
//             compression 2.98 ms, Jubjub 1.44 ms, flat up to 4 096 items (one wavefront per SIMD).
//   LPR = 64  ONE item per wavefront, a_i and the quotient digit through v_readlane -> SGPR -> scalar ALU (rounds
//             1-2's latency kernel): 4.10 / 1.96 ms -- the scalar round trip costs more than the VALU digit, and a
//             wavefront carries a quarter of the items.  Kept selectable (ANEMOI_COOP_MAX) for A/B and parity.
// Cut-offs from the per-size sweeps (tools/sweep_coop.py, profiles/r03/coop_kernel_sweep.txt): the row-cooperative
// kernel wins up to 8 192 items on both limb counts (Jubjub 2.25 vs 2.41 ms, BLS12-381 4.82 vs 7.29 ms at 8 192;
// lane-private from 16 384: 2.41 vs 3.98, 7.27 vs 8.81) -- that is two wavefronts of four items on each of the
// MI355X's 1 024 SIMDs, and the automatic cut-offs are stated that way (`simds` = 4 x the device's CU count, so a
// partitioned or smaller device scales them).  Each is an option (options.h: read from the environment ONCE, changed
// through anemoi_set_option; the parity tests force each kernel for every size that way).
// One item per wavefront (k_jive2_coop<F, 64>): NEVER by default -- A/B and parity only.  On the 11-limb fields this is
// the FOUR-row fold product (an element on all four rows: half the multiply-adds per wavefront, 67 instructions per
// squaring instead of 73), which measures SLOWER than the two-row form (Jubjub 1.073 vs 0.960 ms per compression,
// profiles/r04/coop_kernel_sweep.txt): its two extra swap levels and three-row form builds leave 18 hazard slots per
// product that nothing can fill (86 issue slots against 79), and a lone wavefront pays for every slot.  On the 15-limb
// fields it is rounds 1-2's one-element scan kernel.
#if ANEMOI_AB_BUILD
inline size_t coop_max_items() { return size_t(opt::get_or(opt::kCoopMax, 0)); }
#endif
// Anemoi-2-1 batches up to this size take the two-row fold kernels (coop2d.h, two items per wavefront): one
// wavefront per SIMD -- beyond that the second wavefront of a SIMD costs more than the scan kernel's extra instructions
inline size_t coop2d_max_items(int simds) { return size_t(opt::get_or(opt::kCoop2dMax, 2ll * simds)); }
inline size_t coop4_max_items(int simds) { return size_t(opt::get_or(opt::kCoop4Max, 8ll * simds)); }
// Anemoi-4-3: batches up to this many states take the row-cooperative kernel k_jive4_coop (two states per wavefront),
// larger ones the lane-pair kernel: two wavefronts per SIMD on every field (profiles/r04/coop_kernel_sweep.txt: 4 096
// states BN-254 1.44 vs 1.57 ms, Jubjub 1.43 vs 1.53, BLS12-381 3.06 vs 4.29; 8 192: 2.65 vs 1.58, 5.74 vs 4.29 -- round
// 3 stopped the 9-limb fields at 2 048, where the two were level before the statements sat on the fetch grid)
inline size_t coop43_max_items(int simds, int /*limbs29*/) {
  return size_t(opt::get_or(opt::kCoop43Max, 4ll * simds));
}
// Anemoi-4-3 batches (Jive, permutation, equal-length sponge) up to this many states take the two-row fold kernels with
// ONE state per wavefront (a column per row pair): one wavefront per SIMD
inline size_t coop2d43_max_items(int simds) { return size_t(opt::get_or(opt::kCoop2d43Max, 1ll * simds)); }
// Sponge batches (whole messages of one length in one launch) of up to this many messages take k_sponge_coop
// (1 KB messages: Jubjub 2-1 49.1 vs 86.5 ms up to 1 024 messages, 51.6 vs 86.7 at 4 096, 134.6 vs 86.8 at 16 384;
// BN-254 4-3 12.0 vs 22.4 ms, 18.3 vs 22.4 at 4 096, 61.8 vs 22.9 at 16 384)
inline size_t coop_sponge_max_items(int simds) { return size_t(opt::get_or(opt::kCoopSpongeMax, 4ll * simds)); }
// the largest equal-length sponge batch ANY cooperative kernel takes (Launch::sponge_seg routes on these three): what
// the host path means by "a latency batch"
inline size_t sponge_latency_max_items(int width, int simds) {
  const size_t fold = width == 2 ? coop2d_max_items(simds) : coop2d43_max_items(simds), scan = coop_sponge_max_items(simds);
  return fold > scan ? fold : scan;
}
// Batches of authentication paths up to this size climb on k_merkle_climb_coop (four paths per wavefront); its own
// knob (round 3 borrowed the sponge's, so that one switched two kernels)
inline size_t coop_climb_max_items(int simds) { return size_t(opt::get_or(opt::kCoopClimbMax, 4ll * simds)); }

// Flystel S-box (src/traits.rs:326-358) on the cooperative arithmetic: x -= g y^2 ; y -= x^(1/alpha) ; x += g y^2 + delta,
// the exponentiation by a sliding window of F::kCoopWin bits over odd powers held in LDS (one word per lane and entry)
template <class F, class C>
__device__ __forceinline__ void coop_flystel(uint32_t& x, uint32_t& y, const typename C::K& k, uint32_t* tab,
                                             const PermConsts& pc) {
  constexpr int E = 1 << (F::kCoopWin - 1);
  const uint32_t lane = threadIdx.x;
  uint32_t t = C::mul(y, y, k);
  x = C::sub(x, C::mul_g(t, k), k);
  {
    const uint32_t x2 = C::mul(x, x, k);
    uint32_t pw = x;
    tab[lane] = pw;
#pragma nounroll
    for (int i = 1; i < E; i++) {
      pw = C::mul(pw, x2, k);
      tab[i * kBlock + lane] = pw;
    }
    uint32_t acc = tab[pc.first5 * kBlock + lane], tmp = acc;
    // A lone wavefront pays every instruction -- and every exposed latency -- between two products, so the step loop
    // is software-pipelined: the schedule word of step s + 2 (scalar load) and the table operand of step s + 1 (LDS,
    // its address from the word fetched a step earlier) are requested BEFORE step s's products and waited for after
    // them; the operand is clamped instead of branched around for the "no multiplication" step; the tmp-register
    // operations of the leading-run doubling exist only for the fields whose schedule uses them (Pallas / Vesta).
    // (The words one and two past the end of the schedule are inside the constants blob.)
    // (volatile: hipcc otherwise re-reads the operand inside the step instead of carrying it across the back-edge, and
    // then waits for it in front of the products)
    typedef const volatile uint32_t __attribute__((address_space(3))) * LdsWords;
    const LdsWords vtab = (LdsWords)(uintptr_t)tab;   // (tab is __shared__: the low 32 bits of the flat address are the LDS offset)
    // operand address = table base + (idx << 8) bytes = word & 0xff00 (an entry is kBlock = 64 words), masked to the
    // table so that the markers 253..255 read a harmless entry instead of being branched around
    static_assert(kBlock == 64, "idx << 8 is the byte offset of table entry idx");
    auto operand_of = [&](uint32_t w) { return vtab[((w & (uint32_t(E - 1) << 8)) >> 2) + lane]; };
    uint32_t word = uniform_word(pc.sched5, 0), next = uniform_word(pc.sched5, 1);
    uint32_t opnd = operand_of(word);
    // one general step: decode, the tmp-register operations of the leading-run doubling, squarings and / or a multiplication
    auto general_step = [&](uint32_t w, uint32_t b) {
      const uint32_t nsq = w & 0xff, idx = w >> 8;
      if constexpr (F::kChainTmp) {
        if (idx == 253) {  // leading-run doubling, see sliding_window() in tools/gen_params.py
          tmp = acc;
          return;
        }
        b = idx == 254 ? tmp : b;
      }
      if (idx == 255) {            // the trailing squarings of the exponent
        if (nsq) acc = C::sqr_n(acc, nsq, k);
      } else if (nsq) {
        acc = C::sqr_mul(acc, nsq, b, k);   // one statement: the operand's lane shifts ride in the last squaring
      } else {
        acc = C::mul(acc, b, k);
      }
    };
    constexpr int kGeneral = F::kCoopRegular ? F::kCoopPrefix : 1 << 30;   // steps that go through the general form
    int s = 0;
#pragma nounroll
    for (; s < pc.steps5 && s < kGeneral; s++) {
      const uint32_t after = uniform_word(pc.sched5, s + 2);
      const uint32_t opnd_next = operand_of(next);
      general_step(word, opnd);
      word = next, next = after, opnd = opnd_next;
    }
    if constexpr (F::kCoopRegular) {
      // the regular tail, every step but the last: squarings, then a table multiplication -- no decoding, no branches
#pragma nounroll
      for (; s < pc.steps5 - 1; s++) {
        const uint32_t after = uniform_word(pc.sched5, s + 2);
        const uint32_t opnd_next = operand_of(next);
        acc = C::sqr_mul(acc, word & 0xff, opnd, k);
        word = next, next = after, opnd = opnd_next;
      }
      if ((word >> 8) == 255) acc = C::sqr_n(acc, word & 0xff, k);   // the trailing squarings of the exponent
      else acc = C::sqr_mul(acc, word & 0xff, opnd, k);
    }
    t = acc;
  }
  y = C::sub(y, t, k);
  t = C::mul(y, y, k);
  x = C::add(C::add(x, C::mul_g(t, k)), k.delta);
}

// the instance's round constants in the limb layout of the arithmetic (PermConsts carries both)
template <class C>
struct CoopArk {
  __device__ static __forceinline__ const uint32_t* c(const PermConsts& pc) { return pc.coop_c; }
  __device__ static __forceinline__ const uint32_t* d(const PermConsts& pc) { return pc.coop_d; }
};
#ifndef ANEMOI_BOUNDS_WALK
template <class F, int ROWS>
struct CoopArk<Coop2d<F, ROWS>> {
  __device__ static __forceinline__ const uint32_t* c(const PermConsts& pc) { return pc.fold_c; }
  __device__ static __forceinline__ const uint32_t* d(const PermConsts& pc) { return pc.fold_d; }
};
#endif

// Anemoi::permutation (src/traits.rs:370-378) on the cooperative arithmetic, for one column (x, y) per element row.
//   W = 2: the state is (x, y); mds_layer arm NUM_COLUMNS = 1 (src/traits.rs:136-142).
//   W = 4: a state's two columns sit on two adjacent 16-lane rows -- or, on the two-row fold arithmetic, on the two row
//          pairs of the wavefront -- (column 0 holds (x0, y0) = (state[0], state[2]), column 1 holds (x1, y1)) -- the lane-pair idea of anemoi_perm.h one level up.  The two S-boxes of a
//          round run side by side; only the linear layer (arm 2, src/traits.rs:143-157) couples the rows, through
//          five cross-row exchanges per round.  Every cross-lane operation (the row exchange, the DPP carries
//          inside add / mul_g) is executed by ALL lanes and the result selected afterwards: under divergent control
//          flow a ds_bpermute reads nothing from inactive lanes.
template <class F, class C, int W>
__device__ __forceinline__ void coop_permutation(uint32_t& x, uint32_t& y, const typename C::K& k, uint32_t* tab,
                                                 const PermConsts& pc) {
  constexpr int NL = C::NL, R = W == 2 ? F::kRounds21 : F::kRounds43;
  constexpr int LPI = C::kLanesPerItem;      // lanes of one column: a 16-lane row (scan) or a row pair (two-row fold)
  const uint32_t j = C::limb(), col = W == 4 ? ((threadIdx.x / LPI) & 1) : 0;
  const bool odd = col != 0;
  auto other = [](uint32_t v) { return (uint32_t)__shfl_xor((int)v, LPI); };   // the same limb of the partner column
  // round constants: the NEXT round's pair is fetched while this round's S-box runs (vector loads, one limb per lane)
  auto konst = [&](const uint32_t* tabk, int r) { return j < NL ? tabk[(r * (W / 2) + int(col)) * NL + j] : 0u; };
  const uint32_t* const ark_c = CoopArk<C>::c(pc);
  const uint32_t* const ark_d = CoopArk<C>::d(pc);
  uint32_t kc = konst(ark_c, 0), kd = konst(ark_d, 0);
#pragma nounroll
  for (int r = 0; r <= R; r++) {
    if (r < R) {  // ark_layer (src/traits.rs:111-125): C[r * c + col], D[r * c + col]
      x = C::add(x, kc);
      y = C::add(y, kd);
      if (r + 1 < R) kc = konst(ark_c, r + 1), kd = konst(ark_d, r + 1);
    }
    if constexpr (W == 2) {
      y = C::add(y, x);
      x = C::add(x, y);
    } else {
      // s0 += g s1 ; s1 += g s0 ; s3 += g s2 ; s2 += g s3 ; swap(s2, s3) ; s2 += s0 ; s3 += s1 ; s0 += s2 ; s1 += s3
      uint32_t ox = other(x), oy = other(y);
      uint32_t p = odd ? oy : ox;               // even: s1 (the odd row's x); odd: s2 (the even row's y)
      uint32_t t = C::mul_g_settled(p, k);      // (a product: the fold arithmetic's values are not tight enough to scale)
      uint32_t sx = C::add(x, t), sy = C::add(y, t);
      x = odd ? x : sx;                         // even: s0 += g s1
      y = odd ? sy : y;                         // odd:  s3 += g s2
      ox = other(x), oy = other(y);
      p = odd ? ox : oy;                        // odd: the updated s0; even: the updated s3
      t = C::mul_g_settled(p, k);
      sx = C::add(x, t), sy = C::add(y, t);
      x = odd ? sx : x;                         // odd:  s1 += g s0
      y = odd ? y : sy;                         // even: s2 += g s3
      y = other(y);                             // swap(s2, s3)
      y = C::add(y, x);                         // s2 += s0 ; s3 += s1
      x = C::add(x, y);                         // s0 += s2 ; s1 += s3
    }
    x = C::settle(x, k);  // back below 2p
    y = C::settle(y, k);
    if (r == R) break;  // permutation = R rounds + a final mds_layer
    coop_flystel<F, C>(x, y, k, tab, pc);  // sbox_layer (src/traits.rs:326-358)
  }
}

template <int FIELD, int LPR>
__global__ __launch_bounds__(kBlock) void k_jive2_coop(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                       size_t n, PermConsts pc) {
  using F = FieldC<FIELD>;
  using C = typename CoopArith<F, LPR>::type;
  constexpr int NABI = C::NABI, E = 1 << (F::kCoopWin - 1), PER = kBlock / LPR;
  __shared__ uint32_t tab[E * kBlock];
  const uint32_t lane = threadIdx.x, j = C::limb(), row = lane / LPR;
  const typename C::K k = C::load_consts();
  const size_t groups = (n + PER - 1) / PER;   // a wavefront works on PER consecutive items
  for (size_t g = blockIdx.x; g < groups; g += gridDim.x) {
    const size_t want = g * PER + row;
    const bool live = want < n;
    const size_t item = live ? want : n - 1;   // rows beyond the batch redo the last item (all rows run the same code)
    const uint32_t w0 = j < NABI ? in[(item * 2 + 0) * NABI + j] : 0u;
    const uint32_t w1 = j < NABI ? in[(item * 2 + 1) * NABI + j] : 0u;
    const uint32_t e0 = C::from_abi(w0, k), e1 = C::from_abi(w1, k);
    uint32_t x = e0, y = e1;
    coop_permutation<F, C, 2>(x, y, k, tab, pc);
    // Jive feed-forward: state[0] + state[1] + elems[0] + elems[1] (anemoi_2_1/hasher.rs:102)
    const uint32_t s = C::add(C::add(x, y), C::add(e0, e1));
    const uint32_t o = C::to_abi(s, k);
    if (live && C::writer() && j < NABI) out[item * NABI + j] = o;
  }
}

// Anemoi-4-3 Jive on the row-cooperative arithmetic: TWO states per wavefront (coop_permutation<.., 4>).
// K = 2: out[i] = e_i + e_{i+2} + s_i + s_{i+2} is row-local; K = 4: the two rows' sums are added
// (anemoi_4_3/hasher.rs:148-179).
// LPR = 32: ONE state per wavefront on the two-row fold arithmetic, a column per row pair -- the lowest latency.
template <int FIELD, int K, int LPR = 16>
__global__ __launch_bounds__(kBlock) void k_jive4_coop(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                       size_t n, PermConsts pc) {
  using F = FieldC<FIELD>;
  using C = typename CoopArith<F, LPR>::type;
  static_assert(LPR == 16 || LPR == 32, "a column per 16-lane row, or per row pair");
  constexpr int NABI = C::NABI, E = 1 << (F::kCoopWin - 1), PER = kBlock / (2 * LPR);
  __shared__ uint32_t tab[E * kBlock];
  const uint32_t lane = threadIdx.x, j = C::limb(), row = lane / LPR, col = row & 1;
  const bool odd = col != 0;
  const typename C::K k = C::load_consts();
  const size_t groups = (n + PER - 1) / PER;
  for (size_t g = blockIdx.x; g < groups; g += gridDim.x) {
    const size_t want = g * PER + row / 2;
    const bool live = want < n;
    const size_t item = live ? want : n - 1;
    const uint32_t w0 = j < NABI ? in[(item * 4 + col) * NABI + j] : 0u;
    const uint32_t w1 = j < NABI ? in[(item * 4 + 2 + col) * NABI + j] : 0u;
    const uint32_t e0 = C::from_abi(w0, k), e1 = C::from_abi(w1, k);
    uint32_t x = e0, y = e1;
    coop_permutation<F, C, 4>(x, y, k, tab, pc);
    uint32_t s = C::add(C::add(x, y), C::add(e0, e1));   // this column's share of the Jive sum
    const uint32_t os = (uint32_t)__shfl_xor((int)s, LPR);
    if (K == 4) s = C::add(s, os);
    const uint32_t o = C::to_abi(s, k);
    if (K == 2) {
      if (live && C::writer() && j < NABI) out[(item * 2 + col) * NABI + j] = o;
    } else {
      if (live && !odd && C::writer() && j < NABI) out[item * NABI + j] = o;
    }
  }
}

// Anemoi::permutation (src/traits.rs:370-378), in place, on the row-cooperative arithmetic: four states per wavefront
// (W = 2) or two (W = 4) -- the latency form of k_permutation / k_permutation_pair for small batches.
template <int FIELD, int W, int LPR = 16>
__global__ __launch_bounds__(kBlock) void k_permutation_coop(uint32_t* __restrict__ states, size_t n, PermConsts pc) {
  using F = FieldC<FIELD>;
  using C = typename CoopArith<F, LPR>::type;
  static_assert(LPR <= 32 || W == 2, "the 4-3 form puts a state's two columns on two adjacent rows / row pairs");
  constexpr int NABI = C::NABI, E = 1 << (F::kCoopWin - 1), PER = W == 2 ? kBlock / LPR : kBlock / (2 * LPR);
  __shared__ uint32_t tab[E * kBlock];
  const uint32_t lane = threadIdx.x, j = C::limb(), row = lane / LPR, col = W == 4 ? (row & 1) : 0;
  const typename C::K k = C::load_consts();
  const size_t groups = (n + PER - 1) / PER;
  for (size_t g = blockIdx.x; g < groups; g += gridDim.x) {
    const size_t want = g * PER + (W == 2 ? row : row / 2);
    const bool live = want < n;
    const size_t item = live ? want : n - 1;
    uint32_t* sx = states + (item * W + col) * NABI;          // state[col]
    uint32_t* sy = states + (item * W + W / 2 + col) * NABI;  // state[c + col]
    uint32_t x = C::from_abi(j < NABI ? sx[j] : 0u, k), y = C::from_abi(j < NABI ? sy[j] : 0u, k);
    coop_permutation<F, C, W>(x, y, k, tab, pc);
    const uint32_t ox = C::to_abi(x, k), oy = C::to_abi(y, k);
    if (live && C::writer() && j < NABI) {
      sx[j] = ox;
      sy[j] = oy;
    }
  }
}

// Authentication-path verification (k_merkle_climb) on the row-cooperative arithmetic: four paths per wavefront, each
// climbing its `depth` merges one after the other at the cooperative latency -- for a handful of paths.
template <int FIELD, int LPR = 16>
__global__ __launch_bounds__(kBlock) void k_merkle_climb_coop(const uint32_t* __restrict__ leaves,
                                                              const uint64_t* __restrict__ index,
                                                              const uint32_t* __restrict__ paths, unsigned depth, size_t n,
                                                              uint32_t* __restrict__ out, PermConsts pc) {
  using F = FieldC<FIELD>;
  using C = typename CoopArith<F, LPR>::type;
  constexpr int NABI = C::NABI, E = 1 << (F::kCoopWin - 1), PER = kBlock / LPR;
  __shared__ uint32_t tab[E * kBlock];
  const uint32_t lane = threadIdx.x, j = C::limb(), row = lane / LPR;
  const typename C::K k = C::load_consts();
  const size_t groups = (n + PER - 1) / PER;
  for (size_t g = blockIdx.x; g < groups; g += gridDim.x) {
    const size_t want = g * PER + row;
    const bool live = want < n;
    const size_t item = live ? want : n - 1;
    const uint64_t idx = index[item];
    uint32_t cur = C::from_abi(j < NABI ? leaves[item * NABI + j] : 0u, k);
#pragma nounroll
    for (unsigned l = 0; l < depth; l++) {
      const uint32_t sib = C::from_abi(j < NABI ? paths[(item * depth + l) * NABI + j] : 0u, k);
      const bool right = (idx >> l) & 1;   // this node is the right child: state = [sibling, node]
      uint32_t x = right ? sib : cur, y = right ? cur : sib;
      const uint32_t sum = C::add(cur, sib);
      coop_permutation<F, C, 2>(x, y, k, tab, pc);
      cur = C::settle(C::add(C::add(x, y), sum), k);   // Jive feed-forward, back below 2p for the next level
    }
    const uint32_t o = C::to_abi(cur, k);
    if (live && C::writer() && j < NABI) out[item * NABI + j] = o;
  }
}

// Lane j's limb of one chunk of a byte message: the chunk's `clen` bytes (clen <= F::kChunk), little-endian, a 0x01 byte
// appended when it is short (hasher.rs:36-57) -- the limb is cut out of the five bytes that hold it.  Reads chunk[0 .. clen).
template <class F, class C>
__device__ __forceinline__ uint32_t coop_chunk_limb(const uint8_t* __restrict__ chunk, int clen, uint32_t j) {
  const int bit = C::W * int(j), b0 = bit >> 3, sh = bit & 7;
  uint64_t v = 0;
#pragma unroll
  for (int t = 0; t < 5; t++) {
    const int at = b0 + t;
    uint32_t byte = 0;
    if (j < uint32_t(C::NL) && at < F::kChunk) {
      if (at < clen) byte = chunk[at];
      else if (at == clen) byte = 1;   // only reachable when clen < kChunk
    }
    v |= uint64_t(byte) << (8 * t);
  }
  return j < uint32_t(C::NL) ? uint32_t(v >> sh) & C::MASK : 0u;
}

// Sponge::hash / hash_field (anemoi_2_1/hasher.rs:18-85, anemoi_4_3/hasher.rs:19-129) on the row-cooperative
// arithmetic, for SMALL batches of messages of equal length: four messages per wavefront (W = 2), two (W = 4).  The
// same unified rule as k_sponge: absorb into state[pos]; permute when pos == RATE or at the last element; when
// num % RATE != 0 a final element 1 is absorbed; digest = state[0].  W = 4: state[0], state[1] = x of the even / odd
// row, state[2] = y of the even row; both rows of a pair decode the same element and the row that owns state[pos]
// keeps the sum.  A byte message's element e is its chunk e (F::kChunk bytes, little-endian, a 0x01 byte appended to
// a short last chunk, hasher.rs:36-57): lane j cuts its 29-bit limb out of the five bytes that hold it.
// One launch absorbs one SEGMENT of every message (SpongeSeg, anemoi_kernels.h; the whole message when first = last = 1):
// a small batch of LONG messages from host memory is fed segment by segment (capi.hip sponge_segments) and keeps the
// latency kernel -- round 4 sent every segmented batch to the lane-private kernels, which at a few thousand messages
// run one wavefront per 16 SIMDs (12.0 against 22.4 ms per 1 KB at 1 024 messages).  The state travels between the
// launches as canonical ABI elements, exactly as k_sponge's does.
template <int FIELD, int W, bool BYTES, int LPR = 16>
__global__ __launch_bounds__(kBlock) void k_sponge_coop(const void* __restrict__ src, size_t per_msg, size_t n,
                                                        uint32_t* __restrict__ out, PermConsts pc, SpongeSeg seg) {
  using F = FieldC<FIELD>;
  using C = typename CoopArith<F, LPR>::type;
  static_assert(LPR <= 32 || W == 2, "the 4-3 form puts a state's two columns on two adjacent rows / row pairs");
  constexpr int NABI = C::NABI, E = 1 << (F::kCoopWin - 1), PER = W == 2 ? kBlock / LPR : kBlock / (2 * LPR), RATE = W - 1;
  __shared__ uint32_t tab[E * kBlock];
  const uint32_t lane = threadIdx.x, j = C::limb(), row = lane / LPR, col = W == 4 ? (row & 1) : 0;
  const bool odd = col != 0;
  const typename C::K k = C::load_consts();
  // per_msg = this launch's bytes / elements per message (the stride of `src`); seg.total_len = the whole message
  const size_t num = BYTES ? (seg.total_len + F::kChunk - 1) / F::kChunk : seg.total_len;
  const size_t total = num + (num % RATE == 0 ? 0 : 1);
  const size_t e_end = seg.last ? total : seg.e0 + (BYTES ? per_msg / F::kChunk : per_msg);
  const size_t groups = (n + PER - 1) / PER;
  for (size_t g = blockIdx.x; g < groups; g += gridDim.x) {
    const size_t want = g * PER + (W == 2 ? row : row / 2);
    const bool live = want < n;
    const size_t item = live ? want : n - 1;
    const uint8_t* msg = (const uint8_t*)src + (BYTES ? item * per_msg : item * per_msg * NABI * 4);
    // this column's part of the state: x = state[col], y = state[W/2 + col]
    uint32_t* const sx_w = seg.state + (item * W + col) * NABI;
    uint32_t* const sy_w = seg.state + (item * W + W / 2 + col) * NABI;
    uint32_t x = 0, y = 0;
    if (!seg.first) {
      x = C::from_abi(j < NABI ? sx_w[j] : 0u, k);
      y = C::from_abi(j < NABI ? sy_w[j] : 0u, k);
    }
    int pos = 0;
#pragma nounroll
    for (size_t e = seg.e0; e < e_end; e++) {
      uint32_t el;
      if (e >= num) {
        el = k.one;
      } else if (BYTES) {
        const size_t c0 = (e - seg.e0) * F::kChunk, left = seg.total_len - e * F::kChunk;
        const int clen = left < size_t(F::kChunk) ? int(left) : F::kChunk;
        el = C::to_mont(coop_chunk_limb<F, C>(msg + c0, clen, j), k);   // plain integer < p -> Montgomery form
      } else {
        const uint32_t w = j < NABI ? ((const uint32_t*)msg)[(e - seg.e0) * NABI + j] : 0u;
        el = C::from_abi(w, k);
      }
      // pos is wave-uniform (every message has the same length)
      const uint32_t sx = C::add(x, el), sy = C::add(y, el);
      if (W == 2) {
        x = sx;
      } else {
        if (pos < 2) x = (odd == (pos == 1)) ? sx : x;
        else y = odd ? y : sy;
      }
      pos++;
      if (pos == RATE || e == total - 1) {
        coop_permutation<F, C, W>(x, y, k, tab, pc);
        pos = 0;
      }
    }
    if (!seg.last) {   // carry the state to the next segment's launch
      const uint32_t ox = C::to_abi(x, k), oy = C::to_abi(y, k);
      if (live && C::writer() && j < NABI) {
        sx_w[j] = ox;
        sy_w[j] = oy;
      }
      continue;
    }
    const uint32_t o = C::to_abi(x, k);   // digest = state[0]
    if (live && !odd && C::writer() && j < NABI) out[item * NABI + j] = o;
  }
}

// The same for messages of DIFFERENT lengths (message i = bytes [off[i], off[i + 1]) of msgs): a SMALL ragged batch -- a
// service's handful of requests -- on the latency kernels instead of the lane-private ragged kernels (one permutation
// 0.96 against 2.25 ms on Jubjub).  The wavefront runs as many steps as its longest message has elements; a message that
// has ended keeps its state (selects, no divergent control flow: the cooperative products run under a full EXEC mask
// whatever the lengths are).  Every live message is at the same position of its rate block (pos = e mod RATE), so the
// only permutation that is not shared is the one behind a message's LAST element: it runs for the whole wavefront and
// the other messages, if in mid-block, drop its result.  `order` and BYTES as in k_sponge_ragged (slot -> message; byte
// messages with byte offsets, or messages of ABI elements with offsets in elements).
template <int FIELD, int W, bool BYTES, int LPR = 16>
__global__ __launch_bounds__(kBlock) void k_sponge_ragged_coop(const uint8_t* __restrict__ msgs, const uint64_t* __restrict__ off,
                                                               size_t n, uint32_t* __restrict__ out, PermConsts pc,
                                                               const uint32_t* __restrict__ order,
                                                               const uint32_t* __restrict__ status) {
  using F = FieldC<FIELD>;
  using C = typename CoopArith<F, LPR>::type;
  static_assert(LPR <= 32, "two or four rows per wavefront");
  constexpr int NABI = C::NABI, E = 1 << (F::kCoopWin - 1), PER = W == 2 ? kBlock / LPR : kBlock / (2 * LPR), RATE = W - 1;
  __shared__ uint32_t tab[E * kBlock];
  const uint32_t lane = threadIdx.x, j = C::limb(), row = lane / LPR, col = W == 4 ? (row & 1) : 0;
  const bool odd = col != 0;
  const typename C::K k = C::load_consts();
  const size_t groups = (n + PER - 1) / PER;
  for (size_t g = blockIdx.x; g < groups; g += gridDim.x) {
    const size_t want = g * PER + (W == 2 ? row : row / 2);
    const bool live = want < n;
    const size_t slot = live ? want : n - 1;
    const size_t item = order ? size_t(order[slot]) : slot;
    if (ragged_refused(status)) {   // malformed offsets (k_sponge_ragged): zero digests, no message byte read; wave-uniform
      if (live && !odd && C::writer() && j < NABI) out[item * NABI + j] = 0u;
      continue;
    }
    const uint64_t o0 = off[item], len = ragged_len(off, item);   // (a decreasing pair reads as an empty message)
    const uint8_t* msg = msgs + o0 * (BYTES ? 1 : NABI * 4);
    const uint64_t num = BYTES ? len / F::kChunk + (len % F::kChunk ? 1 : 0) : len;
    const uint64_t tot = num + (num % RATE == 0 ? 0 : 1);   // + the padding element 1 (never for RATE = 1); 64-bit like the lengths
    const uint64_t steps = wave_max(tot);
    uint32_t x = 0, y = 0;   // this column's part of the state: x = state[col], y = state[W/2 + col]
#pragma nounroll
    for (uint64_t e = 0; e < steps; e++) {
      const bool active = e < tot, have = e < num;
      uint32_t el;
      if (BYTES) {
        const size_t c0 = have ? size_t(e) * F::kChunk : 0;
        const int clen = !have ? 0 : len - c0 < uint64_t(F::kChunk) ? int(len - c0) : F::kChunk;   // (nothing is read when clen = 0)
        el = C::to_mont(coop_chunk_limb<F, C>(msg + c0, clen, j), k);
      } else {
        el = C::from_abi(have && j < NABI ? ((const uint32_t*)msg)[size_t(e) * NABI + j] : 0u, k);
      }
      el = have ? el : k.one;
      const int pos = int(e % RATE);   // wave-uniform: every message still running is at the same place of its block
      const uint32_t sx = C::add(x, el), sy = C::add(y, el);
      uint32_t nx = x, ny = y;
      if (W == 2) {
        nx = sx;
      } else {
        if (pos < 2) nx = (odd == (pos == 1)) ? sx : x;
        else ny = odd ? y : sy;
      }
      x = active ? nx : x;
      y = active ? ny : y;
      const bool need = active && (pos + 1 == RATE || e + 1 == tot);
      if (wave_max(need ? 1u : 0u)) {
        uint32_t px = x, py = y;
        coop_permutation<F, C, W>(px, py, k, tab, pc);
        x = need ? px : x;
        y = need ? py : y;
      }
    }
    const uint32_t o = C::to_abi(x, k);   // digest = state[0]
    if (live && !odd && C::writer() && j < NABI) out[item * NABI + j] = o;
  }
}

}  // namespace anemoi
