// wall_rates.hip -- VALU throughput by WALL CLOCK (hipEvents), not s_memtime: wave-instructions per
// second per SIMD for v_mad_u64_u32 / v_add_u32 / mixes at 1..8 waves per SIMD, reported as
// nanoseconds and as cycles at the clock given on the command line (default 2.36 GHz, the value
// rocprofv3's GRBM_GUI_ACTIVE gave for the real kernel).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define S(i) #i
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define MAD(i) "v_mad_u64_u32 %" S(i) ", vcc, %8, %9, %" S(i) "\n\t"
#define ADD(i) "v_add_u32 %" S(i) ", %" S(i) ", %8\n\t"
#define LSHLADD(i) "v_lshl_add_u64 %" S(i) ", %" S(i) ", 1, %" S(i) "\n\t"
#define FMA64(i) "v_fma_f64 %" S(i) ", %" S(i) ", %" S(i) ", %" S(i) "\n\t"
#define MUL24(i) "v_mul_u32_u24 %" S(i) ", %" S(i) ", %8\n\t"
#define MULLO(i) "v_mul_lo_u32 %" S(i) ", %" S(i) ", %8\n\t"
#define KERNEL(NAME, T, STR)                                                                          \
  __global__ void NAME(T* out, uint32_t x, uint32_t y, int iters) {                                   \
    x += threadIdx.x; y ^= threadIdx.x * 2654435761u;                                                  \
    T r0 = x, r1 = x + 1, r2 = x + 2, r3 = x + 3, r4 = x + 4, r5 = x + 5, r6 = x + 6, r7 = x + 7;     \
    _Pragma("nounroll") for (int i = 0; i < iters; i++)                                               \
      asm volatile(STR STR STR STR : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5),      \
                   "+v"(r6), "+v"(r7) : "v"(x), "v"(y) : "vcc");                                      \
    if (x == 0xdeadbeefu) out[threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                   \
  }
KERNEL(k_mad, uint64_t, R8(MAD))
KERNEL(k_add, uint32_t, R8(ADD))
KERNEL(k_lshladd, uint64_t, R8(LSHLADD))
KERNEL(k_fma64, double, R8(FMA64))
KERNEL(k_mul24, uint32_t, R8(MUL24))
KERNEL(k_mullo, uint32_t, R8(MULLO))
template <class T>
static void run(const char* name, void (*fn)(T*, uint32_t, uint32_t, int), double ghz) {
  T* d;
  (void)hipMalloc(&d, 4096);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  printf("%-18s", name);
  const int iters = 8192;
  for (int wps : {1, 2, 3, 4, 5, 6, 8}) {
    const int grid = 256 * 4 * wps;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64), 0, 0, d, 1u, 2u, 64);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64), 0, 0, d, 1u, 2u, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    // every SIMD executes wps waves x iters x 32 instructions
    double ns_per_instr = ms * 1e6 / ((double)wps * iters * 32);
    printf("  %5.2f", ns_per_instr * ghz);
  }
  printf("\n");
}
int main(int argc, char** argv) {
  double ghz = argc > 1 ? atof(argv[1]) : 2.36;
  printf("cycles (at %.2f GHz) per wave-instruction per SIMD, by wall clock\n%-18s", ghz, "waves/SIMD ->");
  for (int w : {1, 2, 3, 4, 5, 6, 8}) printf("  %5d", w);
  printf("\n");
  run<uint64_t>("v_mad_u64_u32", k_mad, ghz);
  run<uint32_t>("v_add_u32", k_add, ghz);
  run<uint64_t>("v_lshl_add_u64", k_lshladd, ghz);
  run<double>("v_fma_f64", k_fma64, ghz);
  run<uint32_t>("v_mul_u32_u24", k_mul24, ghz);
  run<uint32_t>("v_mul_lo_u32", k_mullo, ghz);
  printf("status: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
