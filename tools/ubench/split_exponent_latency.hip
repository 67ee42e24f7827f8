// split_exponent_latency.hip -- PROTOTYPE of the one latency lever left: the S-box's x^(1/alpha) with the
// squarings on one wavefront and the multiplications on a second one that trails it.
//
// A lone wavefront pays per ISSUE SLOT, so a compression's latency is the slot count of its dependent products.  The shipped
// latency kernel runs a left-to-right sliding window: bits - 1 squarings WITH the window multiplications in between, all on the
// critical path.  Right to left, the squarings x^(2^i) are one dependent chain and the multiplications only CONSUME them:
//   wavefront 0 (squarer)     v = x;  for every set bit i of the exponent, low to high: publish v = x^(2^i) in LDS
//   wavefront 1 (multiplier)  acc = the first published power;  acc = acc * power_j  as the powers arrive
// The critical path is the squarings + one product + the hand-offs.  Both on the two-row fold arithmetic of coop2d.h (two
// elements per wavefront, a value = one VGPR per lane), Jubjub; `rounds` exponentiations chained (the result is the next x, as
// the S-boxes of consecutive rounds are), ONE workgroup on an idle chip.
//   mode 0: one wavefront, left-to-right sliding window (5-bit, table in LDS) -- what the shipped kernel's schedule costs
//   mode 1: two wavefronts, right to left, powers handed over through LDS
// Both results are compared with each other and with the library's anemoi_exp_alpha_batch.
//   hipcc --offload-arch=gfx950 -O3 -I anemoi-rust_amd/csrc -I include tools/ubench/split_exponent_latency.hip \
//         -L anemoi-rust_amd/lib -lanemoi_mi355x -Wl,-rpath,$PWD/anemoi-rust_amd/lib -o tools/ubench/bin/split_exponent_latency
//   tools/ubench/bin/split_exponent_latency <1/alpha mod p-1 in hex> [rounds=21]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "anemoi_mi355x.h"
#include "coop2d.h"
using namespace anemoi;

constexpr int FIELD = 4;   // jubjub
using F = FieldC<FIELD>;
using C = Coop2d<F, 2>;
constexpr int kMaxSteps = 320;

struct Sched {
  int n;                       // mode 0: steps (nsq, table index or 255); mode 1: published powers
  uint8_t a[kMaxSteps];        // mode 0: squarings of the step;            mode 1: squarings BEFORE publishing power j
  uint8_t b[kMaxSteps];        // mode 0: table index (255 = none)
  int first;                   // mode 0: table index the accumulator starts from
};

// bounded: a hand-off that never comes sets the error word, and every later wait returns at once (the kernel always ends)
__device__ __forceinline__ void lds_wait_counter(volatile uint32_t* c, uint32_t want, volatile uint32_t* err) {
  for (int it = 0; it < 400000; it++) {
    if (*c >= want || *err) return;
    __builtin_amdgcn_s_sleep(1);
  }
  *err = 1;
}

template <int MODE>
__global__ __launch_bounds__(128) void k(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t* __restrict__ ticks,
                                         Sched s, int rounds) {
  __shared__ uint32_t tab[136 * 64];          // mode 0: 16 table entries; mode 1: up to 130 published powers + the result slot
  __shared__ uint32_t flag[4];
  __shared__ uint32_t runs[136];                // [0] powers published (monotonic over the rounds), [1] results handed back
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = C::limb();
  const typename C::K kk = C::load_consts();
  if (threadIdx.x < 4) flag[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t item = lane >> 5;            // two elements per wavefront
  uint32_t x = wave == 0 ? C::from_abi(j < C::NABI ? in[item * C::NABI + j] : 0u, kk) : 0u;
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  if (MODE == 2 || MODE == 3) {   // what breaking a run of squarings costs: 253 squarings as ONE statement / as 253 statements
    if (wave == 0) {
      for (int r = 0; r < rounds; r++) {
        if (MODE == 2) x = C::sqr_n(x, 253u, kk);
        else {
#pragma nounroll
          for (int i = 0; i < 253; i++) x = C::sqr_n(x, 1u, kk);
        }
      }
    }
  } else if (MODE == 0) {
    if (wave == 0) {
      for (int r = 0; r < rounds; r++) {
        const uint32_t x2 = C::mul(x, x, kk);
        uint32_t pw = x;
        tab[lane] = pw;
        for (int i = 1; i < 16; i++) {
          pw = C::mul(pw, x2, kk);
          tab[i * 64 + lane] = pw;
        }
        uint32_t acc = tab[s.first * 64 + lane];
        for (int st = 0; st < s.n; st++) {
          const uint32_t nsq = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.a[st]), idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.b[st]);
          if (idx == 255) acc = C::sqr_n(acc, nsq, kk);
          else acc = C::sqr_mul(acc, nsq, tab[idx * 64 + lane], kk);
        }
        x = acc;
      }
    }
  } else {
    volatile uint32_t* pub = &flag[0];
    volatile uint32_t* back = &flag[1];
    volatile uint32_t* err = &flag[2];
    if (wave == 0) {           // the squarer
      // the run lengths wait in LDS (one word per power) and the NEXT one is requested before the squarings of this one, so
      // that its latency hides behind them (the shipped kernels software-pipeline their schedule words the same way)
      for (int p = lane; p < s.n; p += 64) runs[p] = s.a[p];
      __builtin_amdgcn_s_waitcnt(0xc07f);
      for (int r = 0; r < rounds; r++) {
        uint32_t v = x;
        tab[lane] = v;                                  // power 0: x itself (the exponent is odd)
        *pub = uint32_t(r * s.n + 1);
        uint32_t nxt = ((volatile uint32_t*)runs)[1];
#pragma nounroll
        for (int p = 1; p < s.n; p++) {
          const uint32_t nsq = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt);
          nxt = ((volatile uint32_t*)runs)[p + 1 < s.n ? p + 1 : p];
          v = C::sqr_n(v, nsq, kk);
          tab[p * 64 + lane] = v;
          *pub = uint32_t(r * s.n + p + 1);            // every lane writes the same number: no exec-mask juggling; the LDS
                                                       // operations of a wavefront complete in order, so the power is there first
        }
        lds_wait_counter(back, uint32_t(r + 1), err);       // the next x is this round's result
        x = tab[135 * 64 + lane];
      }
    } else {                   // the multiplier
      for (int r = 0; r < rounds; r++) {
        lds_wait_counter(pub, uint32_t(r * s.n + 1), err);
        uint32_t acc = tab[lane];
        for (int p = 1; p < s.n; p++) {
          lds_wait_counter(pub, uint32_t(r * s.n + p + 1), err);
          acc = C::mul(acc, tab[p * 64 + lane], kk);
        }
        tab[135 * 64 + lane] = acc;
        *back = uint32_t(r + 1);
        x = acc;
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
  const bool owner = wave == 0;     // (coop2d.h's conversions index the wavefront by threadIdx.x: wave 0 only; in mode 1 it holds the result too)
  if (owner) {
    const uint32_t o = C::to_abi(x, kk);
    if (C::writer() && j < C::NABI) out[item * C::NABI + j] = o;
    if (lane == 0) ticks[0] = t1 - t0, ticks[1] = flag[2];
  }
  if (lane == 0) {
    ticks[2 + wave] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_ID: which SIMD each wavefront ran on
  }
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const int rounds = argc > 2 ? atoi(argv[2]) : 21;
  // exponent bits, least significant first
  std::string hex = argv[1];
  std::vector<int> bits;
  for (int i = int(hex.size()) - 1; i >= 0; i--) {
    const char c = hex[i];
    const int v = c >= 'a' ? c - 'a' + 10 : c >= 'A' ? c - 'A' + 10 : c - '0';
    for (int b = 0; b < 4; b++) bits.push_back((v >> b) & 1);
  }
  while (!bits.empty() && !bits.back()) bits.pop_back();
  const int nb = int(bits.size());
  Sched s0{}, s1{};
  {   // mode 0: left-to-right sliding window of 5 bits over odd powers x^1 .. x^31 (table index = (value - 1) / 2)
    int i = nb - 1;
    bool started = false;
    int pending = 0;
    while (i >= 0) {
      if (!bits[i]) {
        pending++, i--;
        continue;
      }
      int lo = i - 4 < 0 ? 0 : i - 4;
      while (!bits[lo]) lo++;
      int val = 0;
      for (int t = i; t >= lo; t--) val = val * 2 + bits[t];
      const int len = i - lo + 1;
      if (!started) s0.first = (val - 1) / 2, started = true;
      else s0.a[s0.n] = uint8_t(pending + len), s0.b[s0.n] = uint8_t((val - 1) / 2), s0.n++;
      pending = 0;
      i = lo - 1;
    }
    if (pending) s0.a[s0.n] = uint8_t(pending), s0.b[s0.n] = 255, s0.n++;
  }
  {   // mode 1: right to left -- power j is published after a[j] more squarings
    int gap = 0;
    for (int i = 0; i < nb; i++) {
      if (bits[i]) s1.a[s1.n++] = uint8_t(gap), gap = 0;
      gap++;
    }
  }
  int sq0 = 0, mu0 = 15 + 1;   // table build: x^2 and 15 products
  for (int i = 0; i < s0.n; i++) sq0 += s0.a[i], mu0 += s0.b[i] != 255;
  int sq1 = 0;
  for (int i = 0; i < s1.n; i++) sq1 += s1.a[i];
  printf("jubjub, exponent of %d bits (%d set).  mode 0: %d squarings + %d other products on ONE wavefront;  mode 1: %d squarings on the squarer, %d products on the multiplier\n",
         nb, s1.n, sq0, mu0, sq1, s1.n - 1);
  if (s0.n > kMaxSteps || s1.n > 130) return 3;

  uint64_t hin[2][4], hexp[2][4];
  for (int it = 0; it < 2; it++)
    for (int l = 0; l < 4; l++) hin[it][l] = 0x0123456789abcdefull * (it + 3) + 0x1111111111111111ull * l, hin[it][3] &= 0x0fffffffffffffffull;
  memcpy(hexp, hin, sizeof hin);
  for (int r = 0; r < rounds; r++)
    if (anemoi_exp_alpha_batch(FIELD, 1, &hexp[0][0], 2, 0) != 0) return 4;     // the library's x^(1/alpha), `rounds` times
  uint32_t *din, *dout;
  uint64_t* dt;
  (void)hipMalloc(&din, 64), (void)hipMalloc(&dout, 64), (void)hipMalloc(&dt, 32);
  (void)hipMemcpy(din, hin, 64, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; mode++) {
    double best = 1e30;
    uint64_t got[2][4];
    for (int rep = 0; rep < 5; rep++) {
      (void)hipMemset(dout, 0, 64);
      (void)hipMemset(dt, 0, 32);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(128), 0, 0, din, dout, dt, s0, rounds);
      else hipLaunchKernelGGL(k<1>, dim3(1), dim3(128), 0, 0, din, dout, dt, s1, rounds);
      const hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
      if (e1 != hipSuccess || e2 != hipSuccess) printf("mode %d: launch %s, synchronise %s\n", mode, hipGetErrorString(e1), hipGetErrorString(e2));
      uint64_t tt[4];
      (void)hipMemcpy(tt, dt, 32, hipMemcpyDeviceToHost);
      if (rep == 0) printf("mode %d: wavefront 0 on SIMD %u of CU %u, wavefront 1 on SIMD %u of CU %u\n", mode, unsigned(tt[2] >> 4) & 3, unsigned(tt[2] >> 8) & 15,
                           unsigned(tt[3] >> 4) & 3, unsigned(tt[3] >> 8) & 15);
      const uint64_t t = tt[0];
      if (tt[1]) printf("mode %d: a hand-off timed out\n", mode);
      (void)hipMemcpy(got, dout, 64, hipMemcpyDeviceToHost);
      best = t * 1e-2 < best ? t * 1e-2 : best;     // 100 MHz ticks -> us
    }
    const bool ok = !memcmp(got, hexp, 64);
    printf("mode %d: %8.1f us for %d chained exponentiations = %7.2f us each   %s\n", mode, best, rounds, best / rounds,
           ok ? "== the library's anemoi_exp_alpha_batch" : "MISMATCH against the library");
  }
  for (int mode = 2; mode < 4; mode++) {
    double best = 1e30;
    for (int rep = 0; rep < 5; rep++) {
      (void)hipMemset(dt, 0, 32);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(128), 0, 0, din, dout, dt, s0, rounds);
      else hipLaunchKernelGGL(k<3>, dim3(1), dim3(128), 0, 0, din, dout, dt, s0, rounds);
      (void)hipDeviceSynchronize();
      uint64_t tt[4];
      (void)hipMemcpy(tt, dt, 32, hipMemcpyDeviceToHost);
      best = tt[0] * 1e-2 < best ? tt[0] * 1e-2 : best;
    }
    printf("mode %d: %8.1f us for %d x 253 squarings = %7.2f us per 253 (%s)\n", mode, best, rounds, best / rounds,
           mode == 2 ? "one statement: the squarings alone" : "253 statements of one squaring: every run broken");
  }
  printf("status: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
