// wave_fairness.hip -- do wavefronts that share a SIMD share it fairly?  Every wavefront runs the same chain of
// Montgomery squarings (the lane-private Jubjub squaring of mont29_asm_gen.h: the instruction mix of the real kernels)
// and records when it started and ended (s_memrealtime, 100 MHz) and where it ran (HW_ID: wave slot, SIMD, CU, SE;
// XCC_ID).  Launched with exactly 1, 2, 3, 4 wavefronts per SIMD (1 024 .. 4 096 single-wavefront workgroups), plain
// and with the round-by-round alternating priority of anemoi_perm.h.  Prints, per launch: kernel time, the spread of
// per-wavefront durations (a launch is as slow as its slowest wavefront), mean residency, and the mean duration by wave
// slot parity and by XCD -- which says whether the loss of an underfilled launch is unfair arbitration inside a SIMD or
// speed differences between XCDs.
//   hipcc --offload-arch=gfx950 -O3 -I anemoi-rust_amd/csrc tools/ubench/wave_fairness.hip -o /tmp/wave_fair && /tmp/wave_fair
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "mont29_asm_gen.h"
using namespace anemoi;

struct Rec {
  uint64_t t0, t1;
  uint32_t hw_id, xcc_id;
};

template <int MODE>   // 0 plain, 1 alternate priority every `period` squarings by slot parity
__global__ __launch_bounds__(64) void k(Rec* rec, uint32_t* out, int iters, int period) {
  const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));     // HW_ID, all 32 bits
  const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));   // XCC_ID
  const uint32_t slot = hw & 1u;
  uint32_t a[9];
#pragma unroll
  for (int i = 0; i < 9; i++) a[i] = (threadIdx.x * 2654435761u + i * 40503u + blockIdx.x) & ((1u << 29) - 1);
  a[8] &= 0xffff;
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
    if (MODE == 1 && i % period == 0) {
      if (((i / period) + slot) & 1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    AsmMont<4, 29>::sqr(a);
  }
  const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) rec[blockIdx.x] = Rec{t0, t1, hw, xcc};
  if (iters < 0) out[blockIdx.x * 64 + threadIdx.x] = a[0] ^ a[8];
}

template <int MODE>
static void run(int grid, int iters, int period, const char* label) {
  Rec* d;
  uint32_t* o;
  (void)hipMalloc(&d, size_t(grid) * sizeof(Rec));
  (void)hipMalloc(&o, size_t(grid) * 64 * 4);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, d, o, 2000, period);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, d, o, iters, period);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  std::vector<Rec> h(grid);
  (void)hipMemcpy(h.data(), d, size_t(grid) * sizeof(Rec), hipMemcpyDeviceToHost);
  uint64_t first = ~0ull, last = 0;
  for (auto& r : h) first = std::min(first, r.t0), last = std::max(last, r.t1);
  std::vector<double> dur;
  double by_slot[2] = {0, 0}, by_xcc[8] = {0};
  int n_slot[2] = {0, 0}, n_xcc[8] = {0};
  double late_start = 0;
  for (auto& r : h) {
    const double t = double(r.t1 - r.t0) / 100.0;   // us
    dur.push_back(t);
    by_slot[r.hw_id & 1] += t, n_slot[r.hw_id & 1]++;
    by_xcc[r.xcc_id & 7] += t, n_xcc[r.xcc_id & 7]++;
    late_start = std::max(late_start, double(r.t0 - first) / 100.0);
  }
  std::sort(dur.begin(), dur.end());
  double sum = 0;
  for (double t : dur) sum += t;
  const double span = double(last - first) / 100.0;
  printf("%-9s %4d wavefronts (%.0f per SIMD): kernel %8.3f ms | per-wavefront us: min %8.1f  median %8.1f  max %8.1f | "
         "mean residency %.3f | latest start +%.1f us\n",
         label, grid, grid / 1024.0, ms, dur.front(), dur[dur.size() / 2], dur.back(), sum / dur.size() / span, late_start);
  printf("          mean us by wave-slot parity: even %.1f (%d)  odd %.1f (%d) | by XCD:", by_slot[0] / std::max(n_slot[0], 1),
         n_slot[0], by_slot[1] / std::max(n_slot[1], 1), n_slot[1]);
  for (int x = 0; x < 8; x++) printf(" %.1f", n_xcc[x] ? by_xcc[x] / n_xcc[x] : 0.0);
  printf("\n");
  (void)hipFree(d);
  (void)hipFree(o);
}

int main() {
  const int iters = 30000;   // ~190 instructions each: ~10-25 ms per wavefront
  for (int grid : {1024, 2048, 3072, 4096}) {
    run<0>(grid, iters, 1, "plain");
    run<1>(grid, iters, 300, "alt/300");
    run<1>(grid, iters, 20, "alt/20");
  }
  printf("status: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
