// anemoi_generic.h -- Anemoi instances given by their trait constants at run time.
//
// The reference's `Anemoi` trait (src/traits.rs:36-76) is generic over NUM_COLUMNS, NUM_ROUNDS, ARK_C,
// ARK_D and an optional MDS matrix; it carries hard-coded `mds_layer` arms for 1..6 columns and a
// matrix-vector arm for anything wider (src/traits.rs:136-304), of which the 14 shipped instances
// reach only arms 1 and 2.  These kernels run ANY such instance:
//
//   * one state on NUM_COLUMNS adjacent lanes: lane (group, i) holds column i = (x_i, y_i) =
//     (state[i], state[c + i]).  The c S-boxes of a round -- all of the work -- run side by side;
//   * the linear layer is always evaluated in the reference's matrix form
//         x' = M x ;  y' = M rotate_left(y, 1) ;  y'' = x' + y' ;  x'' = x' + y''
//     with lane i computing row i (operands fetched from the group's lanes with ds_bpermute).  For
//     the hard-coded arms the host passes the matrix the arm applies (capi.hip builtin_mds; that the
//     arms and their matrices agree statement by statement is checked in oracle/anemoi_ref.py and
//     tests/test_oracle.py), so results are bit-identical to the arm;
//   * constants arrive in the C-ABI's Montgomery form and are converted ONCE into the kernels' own form
//     (k_generic_prepare: A::NL limbs in R' Montgomery form, padded to whole uint4) -- per shard by the
//     host-pointer entry points, per anemoi_generic_prepare() handle for the device-pointer ones.  Round 2
//     converted them on every use: c + 2 extra products per lane per round.
//
// Loose bounds (mont29.h; walked for 1 .. 16 columns and ANY matrix of canonical entries in BOUNDS.md, tests/test_bounds_walk.py):
// a row sum is c products of a state element (< 12.4 p after the round constant) by a prepared entry (< 1.02 p), each
// < 1.2 p, so < 17.8 p at 16 columns on Jubjub (H = 70.7: 25 % of R'); it is settled BEFORE the PHT step, the PHT sums
// (< 3.6 p) are settled again for the S-box.  k_jive_cols' sum of up to 2c = 32 settled values (< 32.7 p, 47 % of
// Jubjub's R') goes straight into to_abi's product (A B / H <= 0.26).
//
// Also here: k_exp_alpha, the element-wise x^ALPHA (exp_by_alpha, src/traits.rs:94-104) and
// x^(1/ALPHA) (exp_by_inv_alpha), the pair the reference's `test_alpha` checks against each other.
#pragma once
// included from anemoi_kernels.h (after the kernels, before the launchers)

namespace anemoi {

constexpr int kMaxGenericColumns = 16;

struct GenericConsts {
  const uint32_t* ark_c;  // [rounds][cols] elements in the kernels' internal form (device memory), see below
  const uint32_t* ark_d;
  const uint32_t* mds;    // [cols][cols], row-major
  int cols, rounds;
};

// internal-form constant: A::NL limbs (R' Montgomery form, < 2p) in generic_stride<A>() words, zero padded
template <class A>
constexpr int generic_stride() {
  return (A::NL + 3) / 4 * 4;
}

template <class A>
__device__ __forceinline__ void load_k(typename A::Fe& v, const uint32_t* __restrict__ p) {
  uint32_t w[generic_stride<A>()];
#pragma unroll
  for (int q = 0; q < generic_stride<A>() / 4; q++) {
    const uint4 t = ((const uint4*)p)[q];
    w[4 * q] = t.x, w[4 * q + 1] = t.y, w[4 * q + 2] = t.z, w[4 * q + 3] = t.w;
  }
#pragma unroll
  for (int i = 0; i < A::NL; i++) v.l[i] = w[i];
}

template <class A>
__device__ __forceinline__ void load_abi(typename A::Fe& v, const uint32_t* __restrict__ p) {
  uint32_t w[A::NABI];
#pragma unroll
  for (int q = 0; q < A::NABI / 4; q++) {
    const uint4 t = ((const uint4*)p)[q];
    w[4 * q] = t.x, w[4 * q + 1] = t.y, w[4 * q + 2] = t.z, w[4 * q + 3] = t.w;
  }
  A::from_abi(v, w);
}

template <class A>
__device__ __forceinline__ void store_abi(uint32_t* __restrict__ p, const typename A::Fe& v) {
  uint32_t w[A::NABI];
  A::to_abi(w, v);
#pragma unroll
  for (int q = 0; q < A::NABI / 4; q++) ((uint4*)p)[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}

template <class A>
__device__ __forceinline__ void fe_from_lane(typename A::Fe& out, const typename A::Fe& in, int src_lane) {
#pragma unroll
  for (int i = 0; i < A::NL; i++) out.l[i] = (uint32_t)__shfl((int)in.l[i], src_lane);
}

// Which state and column this lane works on.  kBlock / c groups per workgroup; lanes left over (and
// groups beyond the batch) shadow group 0 so that every shuffle source stays inside a live group.
struct ColsGeom {
  int c, col, base;
  size_t item;
  bool live;
  __device__ ColsGeom(int cols, size_t n) {
    c = cols;
    const int groups = kBlock / c, g = threadIdx.x / c;
    col = threadIdx.x - g * c;
    const size_t first = size_t(blockIdx.x) * groups;
    live = g < groups && first + g < n;
    const int gg = live ? g : 0;
    base = gg * c;
    item = first + gg;
  }
};

// linear layer in matrix form (src/traits.rs:281-304), lane `col` computes row `col`
template <class F, class A>
__device__ __forceinline__ void mds_cols(typename A::Fe& x, typename A::Fe& y, const GenericConsts& gc,
                                         const ColsGeom& geo) {
  typename A::Fe nx, ny, m, p, t;
  A::set_zero(nx);
  A::set_zero(ny);
  const int c = geo.c;
#pragma nounroll
  for (int j = 0; j < c; j++) {
    load_k<A>(m, gc.mds + size_t(geo.col * c + j) * generic_stride<A>());
    fe_from_lane<A>(p, x, geo.base + j);
    A::mul(t, p, m);
    A::add(nx, nx, t);
    fe_from_lane<A>(p, y, geo.base + (j + 1 == c ? 0 : j + 1));  // rotate_left(1) of the y half
    A::mul(t, p, m);
    A::add(ny, ny, t);
  }
  if (A::kLoose) {
    A::settle(nx);
    A::settle(ny);
  }
  A::add(y, nx, ny);  // PHT layer
  A::add(x, nx, y);
  if (A::kLoose) {
    A::settle(x);
    A::settle(y);
  }
}

template <class F, class A, int WIN>
__device__ __forceinline__ void permutation_cols(typename A::Fe& x, typename A::Fe& y, const GenericConsts& gc,
                                                 const ColsGeom& geo, const PermConsts& pc, const LdsTable<A>& tab) {
#pragma nounroll
  for (int r = 0; r < gc.rounds; r++) {
    typename A::Fe k;
    load_k<A>(k, gc.ark_c + size_t(r * geo.c + geo.col) * generic_stride<A>());  // ark_layer, src/traits.rs:111-125
    A::add(x, x, k);
    load_k<A>(k, gc.ark_d + size_t(r * geo.c + geo.col) * generic_stride<A>());
    A::add(y, y, k);
    mds_cols<F, A>(x, y, gc, geo);
    flystel<F, A, WIN>(x, y, pc, tab);
  }
  mds_cols<F, A>(x, y, gc, geo);
}

// ABI elements -> internal-form constants, one element per lane (run once per instance / shard)
template <int FIELD>
__global__ __launch_bounds__(kBlock) void k_generic_prepare(const uint32_t* __restrict__ abi, uint32_t* __restrict__ out,
                                                            size_t count) {
  using A = ArithFor<FIELD>;
  const size_t i = size_t(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= count) return;
  typename A::Fe v;
  load_abi<A>(v, abi + i * A::NABI);
  uint32_t* o = out + i * generic_stride<A>();
#pragma unroll
  for (int l = 0; l < generic_stride<A>(); l++) o[l] = l < A::NL ? v.l[l] : 0u;
}

template <int FIELD>
ANEMOI_KERNEL void k_permutation_cols(uint32_t* __restrict__ states, size_t n, GenericConsts gc, PermConsts pc) {
  using F = FieldC<FIELD>;
  using A = ArithFor<FIELD>;
  constexpr int WIN = KernelCfg<F::N>::WIN;
  extern __shared__ uint4 lds[];
  const ColsGeom geo(gc.cols, n);
  uint32_t* sx = states + (geo.item * 2 * geo.c + geo.col) * A::NABI;
  uint32_t* sy = sx + size_t(geo.c) * A::NABI;
  typename A::Fe x, y;
  load_abi<A>(x, sx);
  load_abi<A>(y, sy);
  permutation_cols<F, A, WIN>(x, y, gc, geo, pc, make_table<A>(lds));
  if (geo.live) {
    store_abi<A>(sx, x);
    store_abi<A>(sy, y);
  }
}

// Jive: out[i] = sum_{j<k} (in + perm(in))[i + cc j], cc = 2c / k outputs per state
// (anemoi_4_3/hasher.rs:162-179 with the state width as a run-time value)
template <int FIELD>
ANEMOI_KERNEL void k_jive_cols(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n, int k,
                               GenericConsts gc, PermConsts pc) {
  using F = FieldC<FIELD>;
  using A = ArithFor<FIELD>;
  constexpr int WIN = KernelCfg<F::N>::WIN;
  extern __shared__ uint4 lds[];
  const ColsGeom geo(gc.cols, n);
  const int c = geo.c, cc = 2 * c / k;
  const uint32_t* sx = in + (geo.item * 2 * c + geo.col) * A::NABI;
  typename A::Fe x, y, vx, vy;
  load_abi<A>(x, sx);
  load_abi<A>(y, sx + size_t(c) * A::NABI);
  vx = x;
  vy = y;
  permutation_cols<F, A, WIN>(x, y, gc, geo, pc, make_table<A>(lds));
  A::add(vx, vx, x);
  A::add(vy, vy, y);
  if (A::kLoose) {
    A::settle(vx);
    A::settle(vy);
  }
  typename A::Fe sum, a, b, t;
  A::set_zero(sum);
#pragma nounroll
  for (int j = 0; j < k; j++) {
    const int s = geo.col + cc * j;  // state index feeding output `col` (meaningful on lanes col < cc)
    const int sl = s % c;
    fe_from_lane<A>(a, vx, geo.base + sl);
    fe_from_lane<A>(b, vy, geo.base + sl);
    fe_select<A>(t, s < c, a, b);
    A::add(sum, sum, t);  // < 2k <= 64 (units of p): to_abi's product brings it below 2
  }
  if (geo.live && geo.col < cc) store_abi<A>(out + (geo.item * cc + geo.col) * A::NABI, sum);
}

// Sponge with RATE_WIDTH = rate over `num` elements per message (anemoi_4_3/hasher.rs:19-129 with run-time
// sizes): state[pos] is x of column pos (pos < c) or y of column pos - c; digest = state[0].
template <int FIELD, bool BYTES>
ANEMOI_KERNEL void k_sponge_cols(const void* __restrict__ src, size_t per_msg, size_t n, uint32_t* __restrict__ out,
                                 int rate, GenericConsts gc, PermConsts pc) {
  using F = FieldC<FIELD>;
  using A = ArithFor<FIELD>;
  constexpr int WIN = KernelCfg<F::N>::WIN;
  extern __shared__ uint4 lds[];
  const ColsGeom geo(gc.cols, n);
  const int c = geo.c;
  const size_t num = BYTES ? (per_msg + F::kChunk - 1) / F::kChunk : per_msg;
  const size_t total = num + (num % rate == 0 ? 0 : 1);
  const uint8_t* msg = (const uint8_t*)src + (BYTES ? geo.item * per_msg : geo.item * per_msg * A::NABI * 4);
  typename A::Fe x, y;
  A::set_zero(x);
  A::set_zero(y);
  int pos = 0;
  const LdsTable<A> tab = make_table<A>(lds);
#pragma nounroll
  for (size_t e = 0; e < total; e++) {
    typename A::Fe el, t;
    sponge_element<F, A, BYTES>(el, msg, e, 0, num, per_msg);
    A::add(t, x, el);
    fe_select<A>(x, pos < c && geo.col == pos, t, x);
    A::add(t, y, el);
    fe_select<A>(y, pos >= c && geo.col == pos - c, t, y);
    pos++;
    if (pos == rate || e == total - 1) {
      permutation_cols<F, A, WIN>(x, y, gc, geo, pc, tab);
      pos = 0;
    }
  }
  if (geo.live && geo.col == 0) store_abi<A>(out + geo.item * A::NABI, x);
}

// INV = false: x^ALPHA by the reference's chain for ALPHA (src/traits.rs:94-104; 5 and 11 are the shipped
// values); INV = true: x^(1/ALPHA).  One element per lane, in place.
template <int FIELD, bool INV>
ANEMOI_KERNEL void k_exp_alpha(uint4* __restrict__ elems, size_t n, PermConsts pc) {
  using F = FieldC<FIELD>;
  using A = ArithFor<FIELD>;
  constexpr int WIN = KernelCfg<F::N>::WIN, PER = A::NABI / 4;
  static_assert(F::kAlpha == 5 || F::kAlpha == 11, "chains below cover the shipped ALPHA values");
  extern __shared__ uint4 lds[];
  const size_t blk0 = size_t(blockIdx.x) * kBlock;
  const int cnt = n - blk0 < size_t(kBlock) ? int(n - blk0) : kBlock;
  block_load<PER>(lds, elems, blk0, cnt);
  typename A::Fe x, r;
  lds_get<A>(lds, threadIdx.x, x);
  __syncthreads();
  if (INV) {
    exp_inv_alpha<F, A, WIN>(r, x, pc, make_table<A>(lds));
  } else {
    A::sqr(r, x);
    A::sqr(r, r);
    A::mul(r, r, x);  // x^5
    if (F::kAlpha == 11) {
      A::sqr(r, r);
      A::mul(r, r, x);  // (x^4 x)^2 x
    }
  }
  __syncthreads();
  lds_put<A>(lds, threadIdx.x, r);
  block_store<PER>(lds, elems, blk0, cnt);
}

inline unsigned cols_grid(size_t n, int cols) {
  const size_t groups = size_t(kBlock / cols);
  return unsigned((n + groups - 1) / groups);
}

}  // namespace anemoi
