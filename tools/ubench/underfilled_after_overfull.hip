// underfilled_after_overfull.hip -- WHERE does the dispatcher put the wavefronts of a launch that does not fill the chip,
// depending on what ran before it?  (round 6; the library-level measurements are profiles/r06/underfilled_launch_placement.txt)
//
// The target: 2 048 single-wavefront workgroups with the footprint of config 3's kernel (122 VGPRs, 9 216 B of LDS), each
// recording its HW_ID (SIMD, CU, SH, SE), XCC_ID and its start / end.  An even placement is 1 024 SIMDs x 2 wavefronts; the
// launch takes what its fullest SIMD takes.  Sequences (every launch on the NULL stream, synchronised in between):
//   1. the target after itself                                         (steady state)
//   2. an OVER-FULL launch of another kernel (16 384 workgroups, 154 VGPRs, 12 288 B of LDS: the headline kernel's footprint),
//      then the target
//   3. the same over-full launch, then a do-nothing launch (1 024 x 64 threads asleep ~20 us), then the target
//   4. an over-full launch of the TARGET'S OWN kernel (16 384 workgroups), then the target   (is it "another kernel" or "over-full"?)
//   5. a launch of the other kernel that exactly fills its capacity (3 072 workgroups), then the target
//   hipcc --offload-arch=gfx950 -O3 -I anemoi-rust_amd/csrc tools/ubench/underfilled_after_overfull.hip -o tools/ubench/bin/underfilled_after_overfull
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
#include "mont29_asm_gen.h"
using namespace anemoi;

struct Rec {
  uint64_t t0, t1;
  uint32_t hw_id, xcc_id;
};

template <int CLAIM>   // CLAIM = 0: 122 VGPRs (9-limb kernels), 1: 154 VGPRs (13-limb kernels)
__global__ __launch_bounds__(64) void k(Rec* rec, uint32_t* out, int iters) {
  extern __shared__ uint32_t lds[];
  const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));     // HW_ID
  const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));   // XCC_ID
  if (CLAIM) asm volatile("v_mov_b32 v153, 0" ::: "v153");
  else asm volatile("v_mov_b32 v121, 0" ::: "v121");
  uint32_t a[9];
#pragma unroll
  for (int i = 0; i < 9; i++) a[i] = (threadIdx.x * 2654435761u + i * 40503u + blockIdx.x) & ((1u << 29) - 1);
  a[8] &= 0xffff;
  lds[threadIdx.x] = a[0];
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) AsmMont<4, 29>::sqr(a);
  const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
  if (rec && threadIdx.x == 0) rec[blockIdx.x] = Rec{t0, t1, hw, xcc};
  if (iters < 0) out[blockIdx.x * 64 + threadIdx.x] = a[0] ^ a[8] ^ lds[63 - threadIdx.x];
}
__global__ void k_sleep(unsigned sleeps) {
  for (unsigned i = 0; i < sleeps; i++) __builtin_amdgcn_s_sleep(127);
}

static Rec* d;
static uint32_t* o;

static float timed(void (*fn)()) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  (void)hipEventRecord(a);
  fn();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms;
}

constexpr int kTarget = 2048, kTargetIters = 12000;
static void target() { hipLaunchKernelGGL(k<0>, dim3(kTarget), dim3(64), 9216, 0, d, o, kTargetIters); }
static void overfull_other() { hipLaunchKernelGGL(k<1>, dim3(16384), dim3(64), 12288, 0, (Rec*)nullptr, o, 1500); }
static void overfull_same() { hipLaunchKernelGGL(k<0>, dim3(16384), dim3(64), 9216, 0, (Rec*)nullptr, o, 1500); }
static void exactly_full_other() { hipLaunchKernelGGL(k<1>, dim3(3072), dim3(64), 12288, 0, (Rec*)nullptr, o, 6000); }
static void do_nothing() { hipLaunchKernelGGL(k_sleep, dim3(1024), dim3(64), 0, 0, 6u); }

static void report(const char* label, float pre_ms, float ms) {
  std::vector<Rec> h(kTarget);
  (void)hipMemcpy(h.data(), d, size_t(kTarget) * sizeof(Rec), hipMemcpyDeviceToHost);
  std::map<uint32_t, int> per_simd, per_cu;
  for (auto& r : h) {
    const uint32_t simd = (r.hw_id >> 4) & 3, cu = (r.hw_id >> 8) & 15, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7;
    const uint32_t cu_key = ((r.xcc_id & 7) << 12) | (se << 8) | (sh << 4) | cu;
    per_cu[cu_key]++;
    per_simd[(cu_key << 2) | simd]++;
  }
  int hist[17] = {0}, cu_hist[40] = {0};
  for (auto& kv : per_simd) hist[std::min(kv.second, 16)]++;
  for (auto& kv : per_cu) cu_hist[std::min(kv.second, 39)]++;
  printf("%-74s before %8.3f ms | target %8.3f ms | SIMDs used %4zu:", label, pre_ms, ms, per_simd.size());
  for (int i = 1; i <= 16; i++)
    if (hist[i]) printf(" %d x %d", hist[i], i);
  printf(" | CUs used %3zu:", per_cu.size());
  for (int i = 1; i < 40; i++)
    if (cu_hist[i]) printf(" %d x %d", cu_hist[i], i);
  printf("\n");
  fflush(stdout);
}

int main() {
  (void)hipMalloc(&d, size_t(1 << 16) * sizeof(Rec));
  (void)hipMalloc(&o, size_t(1 << 16) * 64 * 4);
  (void)hipDeviceSynchronize();
  printf("target: %d single-wavefront workgroups (122 VGPRs, 9 216 B LDS), %d squarings each; SIMDs x wavefronts held, CUs x workgroups held\n", kTarget, kTargetIters);
  for (int round = 0; round < 2; round++) {
    (void)timed(target);
    report("1. the target after itself", 0, timed(target));
    float p = timed(overfull_other);
    report("2. after an over-full launch of ANOTHER kernel (16 384 workgroups)", p, timed(target));
    p = timed(overfull_other);
    p += timed(do_nothing);
    report("3. the same, then a do-nothing launch (1 024 x 64 threads, ~20 us)", p, timed(target));
    p = timed(overfull_same);
    report("4. after an over-full launch of the target's OWN kernel (16 384 workgroups)", p, timed(target));
    p = timed(exactly_full_other);
    report("5. after a launch of the other kernel that exactly fills its capacity (3 072)", p, timed(target));
    p = timed(overfull_same);
    p += timed(do_nothing);
    report("6. over-full launch of its own kernel, then the do-nothing launch", p, timed(target));
  }
  printf("status: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
