// vgpr_banks.hip -- does the VGPR bank of a v_mad_u64_u32's source operands matter?  A dependent chain of
//   v_mad_u64_u32 v[100:101], vcc, vX, vY, v[100:101]
// (the accumulator sits in banks 0 and 1: register number mod 4) with the multiplicands in chosen banks, and the same
// with an SGPR multiplicand (the shape of the reduction's m_j * p_k products).  Wall clock by HIP events, 3 wavefronts per
// SIMD (the headline kernel's occupancy) and 1.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/vgpr_banks.hip -o /tmp/vgpr_banks && /tmp/vgpr_banks
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t* out, int iters) {
  uint32_t a = threadIdx.x * 2654435761u + 12345u, b = threadIdx.x * 40503u + 7u;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {   // multiplicands in banks 2, 3: no conflict with the accumulator (banks 0, 1)
      asm volatile("v_mov_b32 v102, %0\n\tv_mov_b32 v103, %1\n\t" REP64("v_mad_u64_u32 v[100:101], vcc, v102, v103, v[100:101]\n\t")
                   : : "v"(a), "v"(b) : "v100", "v101", "v102", "v103", "vcc");
    } else if (MODE == 1) {   // both in bank 0 (= the accumulator's low word)
      asm volatile("v_mov_b32 v104, %0\n\tv_mov_b32 v108, %1\n\t" REP64("v_mad_u64_u32 v[100:101], vcc, v104, v108, v[100:101]\n\t")
                   : : "v"(a), "v"(b) : "v100", "v101", "v104", "v108", "vcc");
    } else if (MODE == 2) {   // banks 0 and 1 (one each with the accumulator's words)
      asm volatile("v_mov_b32 v104, %0\n\tv_mov_b32 v105, %1\n\t" REP64("v_mad_u64_u32 v[100:101], vcc, v104, v105, v[100:101]\n\t")
                   : : "v"(a), "v"(b) : "v100", "v101", "v104", "v105", "vcc");
    } else if (MODE == 3) {   // both in bank 2 (conflict with each other only)
      asm volatile("v_mov_b32 v102, %0\n\tv_mov_b32 v106, %1\n\t" REP64("v_mad_u64_u32 v[100:101], vcc, v102, v106, v[100:101]\n\t")
                   : : "v"(a), "v"(b) : "v100", "v101", "v102", "v106", "vcc");
    } else if (MODE == 4) {   // SGPR multiplicand, VGPR in bank 2
      asm volatile("v_mov_b32 v102, %0\n\ts_mov_b32 s60, 0x12345\n\t" REP64("v_mad_u64_u32 v[100:101], vcc, v102, s60, v[100:101]\n\t")
                   : : "v"(a) : "v100", "v101", "v102", "s60", "vcc");
    } else if (MODE == 5) {   // SGPR multiplicand, VGPR in bank 0
      asm volatile("v_mov_b32 v104, %0\n\ts_mov_b32 s60, 0x12345\n\t" REP64("v_mad_u64_u32 v[100:101], vcc, v104, s60, v[100:101]\n\t")
                   : : "v"(a) : "v100", "v101", "v104", "s60", "vcc");
    } else if (MODE == 6) {   // same register twice (the squaring's diagonal term), bank 2
      asm volatile("v_mov_b32 v102, %0\n\t" REP64("v_mad_u64_u32 v[100:101], vcc, v102, v102, v[100:101]\n\t")
                   : : "v"(a) : "v100", "v101", "v102", "vcc");
    } else {                  // addend 0 instead of the accumulator (a column's first product)
      asm volatile("v_mov_b32 v102, %0\n\tv_mov_b32 v103, %1\n\t" REP64("v_mad_u64_u32 v[100:101], vcc, v102, v103, 0\n\t")
                   : : "v"(a), "v"(b) : "v100", "v101", "v102", "v103", "vcc");
    }
  }
  if (iters < 0) out[threadIdx.x] = a;
}

template <int MODE>
static double run(int grid, int iters) {
  uint32_t* d;
  (void)hipMalloc(&d, 4096);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, d, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, d, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipFree(d);
  // cycles per wave-instruction per SIMD at 2.3 GHz: waves per SIMD * instructions per wave / time
  const double waves_per_simd = grid / 1024.0, instr = double(iters) * 64;
  return ms * 1e-3 * 2.3e9 / (instr * (waves_per_simd < 1 ? 1 : waves_per_simd));
}

int main() {
  const char* names[8] = {"VGPR x VGPR, banks 2,3 (none shared with the accumulator)", "VGPR x VGPR, both bank 0",
                          "VGPR x VGPR, banks 0,1", "VGPR x VGPR, both bank 2", "VGPR(bank 2) x SGPR", "VGPR(bank 0) x SGPR",
                          "same VGPR twice (bank 2)", "VGPR x VGPR, banks 2,3, addend 0"};
  for (int grid : {1024, 3072, 4096}) {
    printf("%d wavefronts (%d per SIMD): cycles per v_mad_u64_u32 per SIMD at 2.3 GHz\n", grid, grid / 1024);
    const int it = 20000;
    double r[8] = {run<0>(grid, it), run<1>(grid, it), run<2>(grid, it), run<3>(grid, it),
                   run<4>(grid, it), run<5>(grid, it), run<6>(grid, it), run<7>(grid, it)};
    for (int m = 0; m < 8; m++) printf("  %-62s %.2f\n", names[m], r[m]);
  }
  printf("status: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
