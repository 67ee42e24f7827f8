// valu_rates.hip -- which multiply instruction should carry the Montgomery multiplier on gfx950?
// Measures issue cost (shader cycles per wave-instruction, from s_memtime) of candidate VALU ops at
// 1, 2 and 4 waves per SIMD, each as 8 independent dependency chains.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define ITER 512
#define BLOCKS_PER_ASM 4  // 4 x 8 = 32 instructions per asm statement

template <class T, class Body>
__device__ void run(unsigned long long* out, T init, Body body) {
  T r0 = init, r1 = init, r2 = init, r3 = init, r4 = init, r5 = init, r6 = init, r7 = init;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma nounroll
  for (int i = 0; i < ITER; i++) body(r0, r1, r2, r3, r4, r5, r6, r7);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("" ::"v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

#define ASM8(STR)                                                                                      \
  asm volatile(STR STR STR STR                                                                         \
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)        \
               : "v"(x), "v"(y), "s"(k)                                                                \
               : "vcc")

#define K64(NAME, STR)                                                                                 \
  __global__ void NAME(unsigned long long* out, uint32_t x, uint32_t y, uint32_t k) {                  \
    x += threadIdx.x; y ^= threadIdx.x * 2654435761u;                                                   \
    run<uint64_t>(out, (uint64_t)x << 7, [&](uint64_t& r0, uint64_t& r1, uint64_t& r2, uint64_t& r3,   \
                                             uint64_t& r4, uint64_t& r5, uint64_t& r6, uint64_t& r7) { \
      ASM8(STR);                                                                                       \
    });                                                                                                \
  }
#define K32(NAME, STR)                                                                                 \
  __global__ void NAME(unsigned long long* out, uint32_t x, uint32_t y, uint32_t k) {                  \
    x += threadIdx.x; y ^= threadIdx.x * 2654435761u;                                                   \
    run<uint32_t>(out, x, [&](uint32_t& r0, uint32_t& r1, uint32_t& r2, uint32_t& r3, uint32_t& r4,    \
                              uint32_t& r5, uint32_t& r6, uint32_t& r7) { ASM8(STR); });               \
  }
#define KF64(NAME, STR)                                                                                \
  __global__ void NAME(unsigned long long* out, uint32_t xi, uint32_t yi, uint32_t k) {                \
    double x = 1.0 + 1e-9 * (xi + threadIdx.x), y = 1.0 - 1e-9 * yi;                                    \
    run<double>(out, x, [&](double& r0, double& r1, double& r2, double& r3, double& r4, double& r5,    \
                            double& r6, double& r7) { ASM8(STR); });                                   \
  }

#define S(i) #i
// one instruction per chain, 8 chains
#define MAD64_VV(i) "v_mad_u64_u32 %" S(i) ", vcc, %8, %9, %" S(i) "\n\t"
#define MAD64_VS(i) "v_mad_u64_u32 %" S(i) ", vcc, %8, %10, %" S(i) "\n\t"
#define LSHLADD64(i) "v_lshl_add_u64 %" S(i) ", %" S(i) ", 1, %" S(i) "\n\t"
#define MULLO(i) "v_mul_lo_u32 %" S(i) ", %" S(i) ", %8\n\t"
#define MULHI(i) "v_mul_hi_u32 %" S(i) ", %" S(i) ", %8\n\t"
#define ADD32(i) "v_add_u32 %" S(i) ", %" S(i) ", %8\n\t"
#define ADD3(i) "v_add3_u32 %" S(i) ", %" S(i) ", %8, %9\n\t"
#define MOV32(i) "v_mov_b32 %" S(i) ", %8\n\t"
#define ADDCO(i) "v_add_co_u32 %" S(i) ", vcc, %" S(i) ", %8\n\t"
#define MAD24(i) "v_mad_u32_u24 %" S(i) ", %" S(i) ", %8, %9\n\t"
#define MUL24(i) "v_mul_u32_u24 %" S(i) ", %" S(i) ", %8\n\t"
#define MULHI24(i) "v_mul_hi_u32_u24 %" S(i) ", %" S(i) ", %8\n\t"
#define MADU16(i) "v_mad_u32_u16 %" S(i) ", %8, %9, %" S(i) "\n\t"
#define DOT2(i) "v_dot2_u32_u16 %" S(i) ", %8, %9, %" S(i) "\n\t"
#define DOT4(i) "v_dot4_u32_u8 %" S(i) ", %8, %9, %" S(i) "\n\t"
#define PKMAD16(i) "v_pk_mad_u16 %" S(i) ", %8, %9, %" S(i) "\n\t"
#define ALIGNBIT(i) "v_alignbit_b32 %" S(i) ", %" S(i) ", %8, 31\n\t"
#define FMA64(i) "v_fma_f64 %" S(i) ", %" S(i) ", %" S(i) ", %" S(i) "\n\t"
#define ADD64F(i) "v_add_f64 %" S(i) ", %" S(i) ", %" S(i) "\n\t"
#define MUL64F(i) "v_mul_f64 %" S(i) ", %" S(i) ", %" S(i) "\n\t"
#define SNOP(i) "s_nop 0\n\t"
// mixes: does a cheap op hide under the multiplier?
#define MAD_ADD(i) "v_mad_u64_u32 %" S(i) ", vcc, %8, %9, %" S(i) "\n\tv_add_u32 %8, %8, %9\n\t"
#define MAD_LSHLADD(i) "v_mad_u64_u32 %" S(i) ", vcc, %8, %9, %" S(i) "\n\tv_lshl_add_u64 %" S(i) ", %" S(i) ", 0, %" S(i) "\n\t"
#define FMA_ADD(i) "v_fma_f64 %" S(i) ", %" S(i) ", %" S(i) ", %" S(i) "\n\tv_lshl_add_u64 %" S(i) ", %" S(i) ", 0, %" S(i) "\n\t"

K64(k_mad64_vv, R8(MAD64_VV))
K64(k_mad64_vs, R8(MAD64_VS))
K64(k_lshladd64, R8(LSHLADD64))
K32(k_mullo, R8(MULLO))
K32(k_mulhi, R8(MULHI))
K32(k_add32, R8(ADD32))
K32(k_add3, R8(ADD3))
K32(k_mov32, R8(MOV32))
K32(k_addco, R8(ADDCO))
K32(k_mad24, R8(MAD24))
K32(k_mul24, R8(MUL24))
K32(k_mulhi24, R8(MULHI24))
K32(k_madu16, R8(MADU16))
K32(k_dot2, R8(DOT2))
K32(k_dot4, R8(DOT4))
K32(k_pkmad16, R8(PKMAD16))
K32(k_alignbit, R8(ALIGNBIT))
KF64(k_fma64, R8(FMA64))
KF64(k_add64f, R8(ADD64F))
KF64(k_mul64f, R8(MUL64F))
K32(k_snop, R8(SNOP))
K64(k_mad_add, R8(MAD_ADD))
K64(k_mad_lshladd, R8(MAD_LSHLADD))
K64(k_fma_lshladd, R8(FMA_ADD))

struct Case {
  const char* name;
  void (*fn)(unsigned long long*, uint32_t, uint32_t, uint32_t);
  int instr_per_rep;  // instructions per chain slot in the string (1 or 2)
};

int main() {
  Case cases[] = {{"v_mad_u64_u32 v,v", k_mad64_vv, 1}, {"v_mad_u64_u32 v,s", k_mad64_vs, 1},
                  {"v_lshl_add_u64", k_lshladd64, 1},   {"v_mul_lo_u32", k_mullo, 1},
                  {"v_mul_hi_u32", k_mulhi, 1},         {"v_add_u32", k_add32, 1},
                  {"v_add3_u32", k_add3, 1},            {"v_mov_b32", k_mov32, 1},
                  {"v_add_co_u32", k_addco, 1},         {"v_mad_u32_u24", k_mad24, 1},
                  {"v_mul_u32_u24", k_mul24, 1},        {"v_mul_hi_u32_u24", k_mulhi24, 1},
                  {"v_mad_u32_u16", k_madu16, 1},       {"v_dot2_u32_u16", k_dot2, 1},
                  {"v_dot4_u32_u8", k_dot4, 1},         {"v_pk_mad_u16", k_pkmad16, 1},
                  {"v_alignbit_b32", k_alignbit, 1},    {"v_fma_f64", k_fma64, 1},
                  {"v_add_f64", k_add64f, 1},           {"v_mul_f64", k_mul64f, 1},
                  {"s_nop 0", k_snop, 1},               {"mad64 + v_add_u32 (pair)", k_mad_add, 2},
                  {"mad64 + lshl_add_u64 (pair)", k_mad_lshladd, 2},
                  {"fma_f64 + lshl_add_u64 (pair)", k_fma_lshladd, 2}};
  unsigned long long* d;
  hipMalloc(&d, sizeof(unsigned long long) * 512 * 16 * 4);
  printf("%-34s %8s %8s %8s %8s %8s %8s  (s_memtime cycles per wave-instruction per SIMD = dt/(n_instr*waves_per_SIMD))\n",
         "instruction", "1 w/SIMD", "2", "3", "4", "6", "8");
  for (auto& c : cases) {
    printf("%-34s", c.name);
    for (int wps : {1, 2, 3, 4, 6, 8}) {
      const int per_cu = wps > 4 ? 2 : 1, block = 256 * wps / per_cu, grid = 256 * per_cu;
      std::vector<unsigned long long> h(grid * block / 64);
      for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(c.fn, dim3(grid), dim3(block), 0, 0, d, 12345u, 6789u, 0x9e3779b9u);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      double med = (double)h[h.size() / 2];
      double n_instr = (double)ITER * BLOCKS_PER_ASM * 8 * c.instr_per_rep;
      printf(" %8.2f", med / n_instr / wps);
    }
    printf("\n");
  }
  hipError_t e = hipGetLastError();
  printf("status: %s\n", hipGetErrorString(e));
  return 0;
}
