// coop_scan_latency.hip -- latency of ONE wave-cooperative Montgomery product, old scan (AsmCoop: the
// quotient digit is computed inside the dependent chain) against the quotient-pipelined scan (AsmCoopQP,
// Orup delay 1: the digit is read one step before it is used).  A chain of dependent products per wave;
// wall clock by HIP events; 1 wavefront alone, then one and two wavefronts on every SIMD.
//   hipcc --offload-arch=gfx950 -O3 -I anemoi-rust_amd/csrc tools/ubench/coop_scan_latency.hip -o /tmp/coop_lat && /tmp/coop_lat
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "mont29_asm_gen.h"
using namespace anemoi;

template <int FIELD, bool QP>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint32_t seed, int iters) {
  const uint32_t lane = threadIdx.x, MASK = (1u << 29) - 1;
  uint32_t a = (seed * 2654435761u + lane * 40503u) & MASK, b = (seed ^ (lane * 97u)) & MASK;
  const uint32_t pl = (0x1234567u + lane) & MASK, sh = lane == 0 ? 29u : 63u;
  if (lane >= 16) a = b = 0;
  for (int i = 0; i < iters; i++) {
    uint64_t t;
    if (QP) t = AsmCoopQP<FIELD>::mul(a, b, pl, pl ^ 5u, sh);
    else t = AsmCoop<FIELD>::mul(a, b, pl, sh);
    a = ((uint32_t)t ^ (uint32_t)(t >> 32)) & MASK;  // next product depends on this one
  }
  if (seed == 0xdeadbeefu) out[blockIdx.x * 64 + lane] = a;
}

template <int FIELD, bool QP>
static double run(int grid, int iters) {
  uint32_t* d;
  (void)hipMalloc(&d, size_t(grid) * 64 * 4);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k<FIELD, QP>), dim3(grid), dim3(64), 0, 0, d, 1u, 16);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k<FIELD, QP>), dim3(grid), dim3(64), 0, 0, d, 1u, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipFree(d);
  return ms * 1e6 / iters;  // ns per product
}

template <int FIELD>
static void field(const char* name) {
  const int iters = 20000;
  for (int grid : {1, 1024, 2048}) {
    const double o = run<FIELD, false>(grid, iters), n = run<FIELD, true>(grid, iters);
    printf("%-10s %5d wavefronts: old scan %7.1f ns / product, quotient-pipelined %7.1f ns  (%.2f x)\n", name, grid, o, n, o / n);
  }
}

int main() {
  field<0>("bls12_381");
  field<4>("jubjub");
  printf("status: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
