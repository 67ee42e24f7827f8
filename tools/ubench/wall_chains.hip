// wall_chains.hip -- does a single wave pay for dependent v_mad_u64_u32 chains?  Wall-clock cycles
// per instruction per wave at 1 / 2 / 3 waves per SIMD with 1, 2, 4, 8 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define S(i) #i
#define MAD(i) "v_mad_u64_u32 %" S(i) ", vcc, %8, %9, %" S(i) "\n\t"
#define C1(X) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0)
#define C2(X) X(0) X(1) X(0) X(1) X(0) X(1) X(0) X(1)
#define C4(X) X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#define C8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, STR)                                                                              \
  __global__ void NAME(uint64_t* out, uint32_t x, uint32_t y, int iters) {                             \
    x += threadIdx.x; y ^= threadIdx.x * 2654435761u;                                                   \
    uint64_t r0 = x, r1 = x + 1, r2 = x + 2, r3 = x + 3, r4 = x + 4, r5 = x + 5, r6 = x + 6, r7 = x + 7; \
    _Pragma("nounroll") for (int i = 0; i < iters; i++)                                                \
      asm volatile(STR STR STR STR : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5),       \
                   "+v"(r6), "+v"(r7) : "v"(x), "v"(y) : "vcc");                                       \
    if (x == 0xdeadbeefu) out[threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                    \
  }
KERNEL(k1, C1(MAD)) KERNEL(k2, C2(MAD)) KERNEL(k4, C4(MAD)) KERNEL(k8, C8(MAD))
int main(int argc, char** argv) {
  double ghz = argc > 1 ? atof(argv[1]) : 2.36;
  uint64_t* d; (void)hipMalloc(&d, 4096);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  void (*fns[])(uint64_t*, uint32_t, uint32_t, int) = {k1, k2, k4, k8};
  const char* names[] = {"1 chain", "2 chains", "4 chains", "8 chains"};
  printf("v_mad_u64_u32: cycles (at %.2f GHz) per instruction per WAVE\n%-10s %8s %8s %8s\n", ghz, "", "1 w/SIMD", "2", "3");
  for (int f = 0; f < 4; f++) {
    printf("%-10s", names[f]);
    for (int wps : {1, 2, 3}) {
      const int grid = 256 * 4 * wps, iters = 8192;
      hipLaunchKernelGGL(fns[f], dim3(grid), dim3(64), 0, 0, d, 1u, 2u, 64);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(a);
      hipLaunchKernelGGL(fns[f], dim3(grid), dim3(64), 0, 0, d, 1u, 2u, iters);
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
      printf(" %8.2f", ms * 1e6 / ((double)iters * 32) * ghz);  // per wave: every wave runs iters*32 instrs
    }
    printf("\n");
  }
  return 0;
}
