// first_launch_placement.hip -- where do the workgroups of the FIRST big launch of a process land?
// profiles/r04/first_launch_after_idle.txt: the first config-3 launch (k_sponge_pair<2, true>: 2 048 single-wavefront
// workgroups = 2 per SIMD, 122 VGPRs, 9 216 B of LDS, ~330 ms) takes x 1.47 in KERNEL time; a >= 4 096-message launch
// of the same kernel first cures it.  x 1.5 is what 3 instead of 2 wavefronts on a SIMD cost, so the question is
// whether the dispatcher spreads the first launch unevenly.  This program launches a kernel of the same shape (same
// VGPR / LDS footprint, 64-thread workgroups, the lane-private Jubjub squaring as its body) and records where every
// workgroup ran (HW_ID: SIMD, CU, SH, SE; XCC_ID), when it started and ended (s_memrealtime, 100 MHz) and its
// shader-clock cycles (s_memtime), for the first, second and third launch of the process:
//   ./first_launch_placement [grid=2048] [iters=60000] [warm=<workgroups of a small first launch>] [idle_ms=<sleep before each launch>]
// Prints per launch: kernel ms, histogram of wavefronts per SIMD, CUs used, workgroups per XCD, the in-kernel clock
// (cycles per 10 ns tick, median over workgroups), the dispatch spread (latest start) and the slowest wavefront.
//   hipcc --offload-arch=gfx950 -O3 -I anemoi-rust_amd/csrc tools/ubench/first_launch_placement.hip -o tools/ubench/first_launch_placement
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <vector>
#include "mont29_asm_gen.h"
using namespace anemoi;

struct Rec {
  uint64_t t0, t1, c0, c1;
  uint32_t hw_id, xcc_id;
};

__global__ __launch_bounds__(64) void k(Rec* rec, uint32_t* out, int iters) {
  extern __shared__ uint32_t lds[];
  const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));     // HW_ID, all 32 bits
  const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));   // XCC_ID
  asm volatile("v_mov_b32 v121, 0" ::: "v121");                                // claim 122 VGPRs like k_sponge_pair<2, true>
  uint32_t a[9];
#pragma unroll
  for (int i = 0; i < 9; i++) a[i] = (threadIdx.x * 2654435761u + i * 40503u + blockIdx.x) & ((1u << 29) - 1);
  a[8] &= 0xffff;
  lds[threadIdx.x] = a[0];
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  const uint64_t c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) AsmMont<4, 29>::sqr(a);
  const uint64_t c1 = __builtin_amdgcn_s_memtime();
  const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) rec[blockIdx.x] = Rec{t0, t1, c0, c1, hw, xcc};
  if (iters < 0) out[blockIdx.x * 64 + threadIdx.x] = a[0] ^ a[8] ^ lds[63 - threadIdx.x];
}

static void launch(int grid, int iters, const char* label, Rec* d, uint32_t* o) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(grid), dim3(64), 9216, 0, d, o, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  std::vector<Rec> h(grid);
  (void)hipMemcpy(h.data(), d, size_t(grid) * sizeof(Rec), hipMemcpyDeviceToHost);
  uint64_t first = ~0ull, last = 0;
  for (auto& r : h) first = std::min(first, r.t0), last = std::max(last, r.t1);
  std::map<uint32_t, int> per_simd, per_cu;
  int per_xcc[8] = {0};
  std::vector<double> dur, clk;
  double late = 0;
  for (auto& r : h) {
    // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
    const uint32_t simd = (r.hw_id >> 4) & 3, cu = (r.hw_id >> 8) & 15, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7;
    const uint32_t cu_key = ((r.xcc_id & 7) << 12) | (se << 8) | (sh << 4) | cu;
    per_cu[cu_key]++;
    per_simd[(cu_key << 2) | simd]++;
    per_xcc[r.xcc_id & 7]++;
    dur.push_back(double(r.t1 - r.t0) / 100.0);
    clk.push_back(double(r.c1 - r.c0) / double(r.t1 - r.t0) * 0.1);   // GHz
    late = std::max(late, double(r.t0 - first) / 100.0);
  }
  int hist[17] = {0};
  for (auto& kv : per_simd) hist[std::min(kv.second, 16)]++;
  int cu_hist[40] = {0};
  for (auto& kv : per_cu) cu_hist[std::min(kv.second, 39)]++;
  std::sort(dur.begin(), dur.end());
  std::sort(clk.begin(), clk.end());
  printf("%-14s %5d workgroups: kernel %9.3f ms | SIMDs used %4zu, by wavefronts held:", label, grid, ms, per_simd.size());
  for (int i = 1; i <= 16; i++)
    if (hist[i]) printf(" %dx%d", hist[i], i);
  printf(" | CUs used %3zu, by workgroups held:", per_cu.size());
  for (int i = 1; i < 40; i++)
    if (cu_hist[i]) printf(" %dx%d", cu_hist[i], i);
  printf("\n               per XCD:");
  for (int x = 0; x < 8; x++) printf(" %d", per_xcc[x]);
  printf(" | in-kernel clock GHz min %.3f median %.3f max %.3f | wavefront us min %.0f median %.0f max %.0f | latest start +%.1f us | span %.3f ms\n",
         clk.front(), clk[clk.size() / 2], clk.back(), dur.front(), dur[dur.size() / 2], dur.back(), late,
         double(last - first) / 1e5);
  // per XCD: when its last workgroup ended (ms after the first start of the launch) and the median clock of its workgroups
  // -- are workgroups dealt to the XCDs statically (equal counts, the slow XCD ends last) or taken as slots free up?
  printf("               per XCD last end ms / median GHz:");
  for (int x = 0; x < 8; x++) {
    uint64_t e = 0;
    std::vector<double> c;
    for (auto& r : h)
      if (int(r.xcc_id & 7) == x) e = std::max(e, r.t1), c.push_back(double(r.c1 - r.c0) / double(r.t1 - r.t0) * 0.1);
    std::sort(c.begin(), c.end());
    printf(" %.2f/%.3f", c.empty() ? 0.0 : double(e - first) / 1e5, c.empty() ? 0.0 : c[c.size() / 2]);
  }
  printf("\n");
  fflush(stdout);
}

int main(int argc, char** argv) {
  int grid = 2048, iters = 60000, warm = 0, idle_ms = 0;
  for (int i = 1; i < argc; i++) {
    if (!strncmp(argv[i], "grid=", 5)) grid = atoi(argv[i] + 5);
    if (!strncmp(argv[i], "iters=", 6)) iters = atoi(argv[i] + 6);
    if (!strncmp(argv[i], "warm=", 5)) warm = atoi(argv[i] + 5);
    if (!strncmp(argv[i], "idle_ms=", 8)) idle_ms = atoi(argv[i] + 8);
  }
  if (grid < 1 || grid > (1 << 16) || iters < 1 || iters > 400000 || warm < 0 || warm > (1 << 16)) return 2;
  Rec* d;
  uint32_t* o;
  (void)hipMalloc(&d, size_t(1 << 16) * sizeof(Rec));
  (void)hipMalloc(&o, size_t(1 << 16) * 64 * 4);
  (void)hipDeviceSynchronize();
  printf("grid=%d iters=%d warm=%d idle_ms=%d\n", grid, iters, warm, idle_ms);
  if (warm) launch(warm, 600, "warm-up", d, o);
  const char* names[3] = {"first launch", "second launch", "third launch"};
  for (int r = 0; r < 3; r++) {
    if (idle_ms) std::this_thread::sleep_for(std::chrono::milliseconds(idle_ms));
    launch(grid, iters, names[r], d, o);
  }
  printf("status: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
