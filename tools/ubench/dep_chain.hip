// dep_chain.hip -- how much instruction-level parallelism does one wave need?  v_mad_u64_u32 issued
// as C independent dependency chains per wave (C = 1, 2, 4, 8) at 1..8 waves per SIMD.
// Output: s_memtime cycles per wave-instruction per SIMD (lower = higher throughput).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
#define ITER 512
#define S(i) #i
#define MAD(i) "v_mad_u64_u32 %" S(i) ", vcc, %8, %9, %" S(i) "\n\t"
#define ADD(i) "v_add_u32 %" S(i) ", %" S(i) ", %8\n\t"
#define C1(X) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0)
#define C2(X) X(0) X(1) X(0) X(1) X(0) X(1) X(0) X(1)
#define C4(X) X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#define C8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, T, STR)                                                                          \
  __global__ void NAME(unsigned long long* out, uint32_t x, uint32_t y) {                             \
    x += threadIdx.x; y ^= threadIdx.x * 2654435761u;                                                  \
    T r0 = x, r1 = x + 1, r2 = x + 2, r3 = x + 3, r4 = x + 4, r5 = x + 5, r6 = x + 6, r7 = x + 7;     \
    unsigned long long t0, t1;                                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                         \
    _Pragma("nounroll") for (int i = 0; i < ITER; i++)                                                \
      asm volatile(STR STR STR STR : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5),      \
                   "+v"(r6), "+v"(r7) : "v"(x), "v"(y) : "vcc");                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                         \
    asm volatile("" ::"v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7));        \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;    \
  }
KERNEL(mad_c1, uint64_t, C1(MAD))
KERNEL(mad_c2, uint64_t, C2(MAD))
KERNEL(mad_c4, uint64_t, C4(MAD))
KERNEL(mad_c8, uint64_t, C8(MAD))
KERNEL(add_c1, uint32_t, C1(ADD))
KERNEL(add_c2, uint32_t, C2(ADD))
KERNEL(add_c4, uint32_t, C4(ADD))
KERNEL(add_c8, uint32_t, C8(ADD))
struct Case { const char* name; void (*fn)(unsigned long long*, uint32_t, uint32_t); };
int main() {
  Case cases[] = {{"v_mad_u64_u32, 1 chain", mad_c1}, {"v_mad_u64_u32, 2 chains", mad_c2},
                  {"v_mad_u64_u32, 4 chains", mad_c4}, {"v_mad_u64_u32, 8 chains", mad_c8},
                  {"v_add_u32, 1 chain", add_c1},     {"v_add_u32, 2 chains", add_c2},
                  {"v_add_u32, 4 chains", add_c4},     {"v_add_u32, 8 chains", add_c8}};
  unsigned long long* d;
  (void)hipMalloc(&d, sizeof(unsigned long long) * 512 * 16 * 4);
  printf("%-28s %8s %8s %8s %8s %8s %8s %8s\n", "waves/SIMD ->", "1", "2", "3", "4", "5", "6", "8");
  for (auto& c : cases) {
    printf("%-28s", c.name);
    for (int wps : {1, 2, 3, 4, 5, 6, 8}) {
      // wps waves per SIMD = 4*wps waves per CU, as blocks of 64*wps... use 64-thread blocks, 4*wps per CU
      const int block = 64, grid = 256 * 4 * wps;
      std::vector<unsigned long long> h(grid);
      for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(c.fn, dim3(grid), dim3(block), 0, 0, d, 12345u, 6789u);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      printf(" %8.2f", (double)h[h.size() / 2] / (ITER * 32.0) / wps);
    }
    printf("\n");
  }
  printf("status: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
