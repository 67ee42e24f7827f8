// lone_wave_fetch.hip -- what does ONE wavefront on an otherwise idle CU pay per instruction, as a function of the
// instruction's ENCODED SIZE, and what does a taken branch cost (target aligned to a fetch line or not)?
// One workgroup of 64 lanes; s_memtime (core clock) around a loop of straight-line blocks.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lwf tools/ubench/lone_wave_fetch.hip && /tmp/lwf
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define TIMED(NAME, BODY_ASM, PER_ITER)                                                                     \
  __global__ void NAME(unsigned long long* out, uint32_t x, int iters) {                                    \
    uint32_t r0 = x + threadIdx.x, r1 = x + 1, r2 = x + 2, r3 = x + 3, r4 = x + 4, r5 = x + 5, r6 = x + 6,  \
             r7 = x + 7;                                                                                     \
    uint64_t w0 = r0, w1 = r1, w2 = r2, w3 = r3;                                                            \
    uint32_t msk = 0x0fffffff + (x >> 31);                                                                  \
    unsigned long long t0, t1, q0, q1;                                                                      \
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(q0), "=s"(t0)::"memory"); \
    asm volatile("s_mov_b32 s90, %12\n\t"                                                                   \
                 ".p2align 6\n\t"                                                                           \
                 "7:\n\t" BODY_ASM                                                                          \
                 "s_sub_u32 s90, s90, 1\n\ts_cmp_lg_u32 s90, 0\n\ts_cbranch_scc1 7b\n\t"                    \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7),          \
                   "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)                                                   \
                 : "s"(iters), "v"(x), "s"(msk)                                                             \
                 : "s90", "s91", "s92", "s93", "scc", "vcc");                                                             \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(q1)::"memory"); \
    asm volatile("" ::"v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(w0),      \
                 "v"(w1), "v"(w2), "v"(w3));                                                                \
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = q1 - q0; out[2] = PER_ITER; }                        \
  }

// 256 instructions per iteration, 8 independent chains
#define R8(a) a(0) a(1) a(2) a(3) a(4) a(5) a(6) a(7)
#define X32(s) ".rept 32\n\t" s ".endr\n\t"
#define ADD32(i) "v_add_u32_e32 %" #i ", %13, %" #i "\n\t"
#define ADD64(i) "v_add_u32_e64 %" #i ", %13, %" #i "\n\t"
#define ANDLIT(i) "v_and_b32_e32 %" #i ", 0x0ffffff1, %" #i "\n\t"
#define ANDSGPR(i) "v_and_b32_e32 %" #i ", %14, %" #i "\n\t"
#define DPP(i) "v_mov_b32_dpp %" #i ", %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define ADDDPP(i) "v_add_u32_dpp %" #i ", %13, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define NOP(i) "s_nop 0\n\t"
#define SALU(i) "s_add_u32 s91, s91, 1\n\t"
TIMED(k_add32, X32(R8(ADD32)), 256)
TIMED(k_add64, X32(R8(ADD64)), 256)
TIMED(k_andlit, X32(R8(ANDLIT)), 256)
TIMED(k_andsgpr, X32(R8(ANDSGPR)), 256)
TIMED(k_dpp, X32(R8(DPP)), 256)
TIMED(k_adddpp, X32(R8(ADDDPP)), 256)
TIMED(k_nop, X32(R8(NOP)), 256)
TIMED(k_salu, X32(R8(SALU)), 256)
// v_mad_u64_u32: 4 independent 64-bit chains (8 B each)
#define MAD(i) "v_mad_u64_u32 %" #i ", vcc, %13, %13, %" #i "\n\t"
#define R4M MAD(8) MAD(9) MAD(10) MAD(11)
TIMED(k_mad, ".rept 64\n\t" R4M ".endr\n\t", 256)
// mixes as in the fold product: mad, dpp mov, add (e32)
#define MIXA MAD(8) DPP(0) ADD32(1) MAD(9) DPP(2) ADD32(3)
TIMED(k_mix, ".rept 42\n\t" MIXA ".endr\n\t", 252)
#define MIXB MAD(8) ADD32(0) ADD32(1) MAD(9) ADD32(2) ADD32(3)
TIMED(k_mix32, ".rept 42\n\t" MIXB ".endr\n\t", 252)

// the other instructions of the fold product, and dependent forms
#define SWAP16(i) "v_permlane16_swap_b32 %" #i ", %1\n\t"
#define SWAP32(i) "v_permlane32_swap_b32 %" #i ", %1\n\t"
#define BCAST(i) "v_mov_b32_dpp %" #i ", %13 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define ROR(i) "v_mov_b32_dpp %" #i ", %13 row_ror:1 row_mask:0xf bank_mask:0x1 bound_ctrl:1\n\t"
#define LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %9\n\t"
#define ALIGNBIT(i) "v_alignbit_b32 %" #i ", %13, %" #i ", 27\n\t"
#define LSHR(i) "v_lshrrev_b32_e32 %" #i ", 27, %" #i "\n\t"
TIMED(k_swap16, X32(SWAP16(0) SWAP16(2) SWAP16(3) SWAP16(4) SWAP16(5) SWAP16(6) SWAP16(7) SWAP16(0)), 256)
TIMED(k_swap32, X32(SWAP32(0) SWAP32(2) SWAP32(3) SWAP32(4) SWAP32(5) SWAP32(6) SWAP32(7) SWAP32(0)), 256)
TIMED(k_bcast, X32(R8(BCAST)), 256)
TIMED(k_ror, X32(R8(ROR)), 256)
TIMED(k_lshladd64, ".rept 64\n\t" LSHLADD64(8) LSHLADD64(10) LSHLADD64(11) LSHLADD64(8) ".endr\n\t", 256)
TIMED(k_alignbit, X32(R8(ALIGNBIT)), 256)
// dependent chains
TIMED(k_mad_dep, ".rept 256\n\t" MAD(8) ".endr\n\t", 256)
TIMED(k_add_dep, ".rept 256\n\t" ADD32(0) ".endr\n\t", 256)
#define MADPAIR "v_mad_u64_u32 %8, vcc, %13, %13, %8\n\tv_mad_u64_u32 %9, vcc, %13, %13, %9\n\t"
TIMED(k_mad_then_dpp, ".rept 64\n\t" MAD(8) ADD32(1) ADD32(2) "v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" ".endr\n\t", 256)
// an add, two independent instructions, then a DPP read of the add's result (the minimum legal distance)
TIMED(k_add_2_dpp, ".rept 64\n\t" ADD32(0) ADD32(1) ADD32(2) "v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" ".endr\n\t", 256)
TIMED(k_add_nop_dpp, ".rept 64\n\t" ADD32(0) "s_nop 1\n\t" "v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" ADD32(1) ".endr\n\t", 256)
// swap right after the add that produced its operand + 2 slots
TIMED(k_add_2_swap, ".rept 64\n\t" ADD32(0) ADD32(2) ADD32(3) "v_permlane16_swap_b32 %0, %1\n\t" ".endr\n\t", 256)
// mad whose 64-bit result is read by the next instruction's 32-bit halves
TIMED(k_mad_use, ".rept 128\n\t" MAD(8) "v_lshl_add_u64 %9, %8, 0, %9\n\t" ".endr\n\t", 256)

#define VNOP64(i) "v_nop_e64\n\t"
#define VNOP32(i) "v_nop\n\t"
#define SUB64(i) "v_sub_u32_e64 %" #i ", 0, %" #i "\n\t"
TIMED(k_vnop64, X32(R8(VNOP64)), 256)
TIMED(k_vnop32, X32(R8(VNOP32)), 256)
TIMED(k_sub64, X32(R8(SUB64)), 256)
// the scan step's gap: a write, one independent instruction, a filler, the DPP read of the write -- filler = s_nop 0 / v_nop_e64
TIMED(k_gap_snop, ".rept 32\n\t" ADD64(0) MAD(8) "s_nop 0\n\ts_nop 0\n\t" "v_add_u32_dpp %3, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" MAD(9) MAD(10) MAD(11) ".endr\n\t", 256)
TIMED(k_gap_vnop, ".rept 32\n\t" ADD64(0) MAD(8) "v_nop_e64\n\t" "v_add_u32_dpp %3, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" MAD(9) MAD(10) MAD(11) ".endr\n\t", 224)

// VALU and SALU / s_nop interleaved one to one
#define VS_SADD(i) "v_add_u32_e64 %" #i ", %13, %" #i "\n\ts_add_u32 s91, s91, 1\n\ts_nop 0\n\t"
#define VS_SMOVLIT(i) "v_add_u32_e64 %" #i ", %13, %" #i "\n\ts_mov_b32 s91, 0x7fffffff\n\t"
#define VS_SNOP(i) "v_add_u32_e64 %" #i ", %13, %" #i "\n\ts_nop 0\n\ts_nop 0\n\t"
#define VS_SCMP(i) "v_add_u32_e64 %" #i ", %13, %" #i "\n\ts_cmp_eq_u32 s91, 0\n\ts_nop 0\n\t"
TIMED(k_vs_sadd, ".rept 16\n\t" R8(VS_SADD) ".endr\n\t", 384)
TIMED(k_vs_smovlit, ".rept 16\n\t" R8(VS_SMOVLIT) ".endr\n\t", 256)
TIMED(k_vs_snop, ".rept 16\n\t" R8(VS_SNOP) ".endr\n\t", 384)
TIMED(k_vs_scmp, ".rept 16\n\t" R8(VS_SCMP) ".endr\n\t", 384)
// the scan step's filler between a VALU write and the DPP read two slots later
#define GAP(F) ".rept 32\n\t" ADD64(0) MAD(8) F "v_add_u32_dpp %3, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" MAD(9) MAD(10) MAD(11) ".endr\n\t"
TIMED(k_gap_smov, GAP("s_mov_b32 s91, 0x7fffffff\n\t"), 224)
TIMED(k_gap_snop0, GAP("s_nop 0\n\t"), 224)

// 64-bit shifts and multiplies of the lane-private product
#define LSHR64(i) "v_lshrrev_b64 %" #i ", 30, %" #i "\n\t"
#define MULLO(i) "v_mul_lo_u32 %" #i ", %13, %" #i "\n\t"
TIMED(k_lshr64, ".rept 64\n\t" LSHR64(8) LSHR64(9) LSHR64(10) LSHR64(11) ".endr\n\t", 256)
TIMED(k_mullo, X32(R8(MULLO)), 256)
// a column of the lane-private squaring: multiply-adds, then the carry shift, dependent
TIMED(k_column, ".rept 32\n\t" MAD(8) MAD(8) MAD(8) MAD(8) MAD(8) MAD(8) MAD(8) LSHR64(8) ".endr\n\t", 256)

// Does a wavefront with only half (or a quarter) of its lanes enabled issue faster?  (EXEC = low 32 / low 16 lanes)
TIMED(k_exec32, "s_mov_b64 s[92:93], exec\n\ts_mov_b64 exec, 0xffffffff\n\t" X32(R8(ADD64)) "s_mov_b64 exec, s[92:93]\n\t", 256)
TIMED(k_exec16, "s_mov_b64 s[92:93], exec\n\ts_mov_b64 exec, 0xffff\n\t" X32(R8(ADD64)) "s_mov_b64 exec, s[92:93]\n\t", 256)
TIMED(k_exec32_mad, "s_mov_b64 s[92:93], exec\n\ts_mov_b64 exec, 0xffffffff\n\t" ".rept 64\n\t" R4M ".endr\n\t" "s_mov_b64 exec, s[92:93]\n\t", 256)
TIMED(k_exec32_dpp, "s_mov_b64 s[92:93], exec\n\ts_mov_b64 exec, 0xffffffff\n\t" X32(R8(DPP)) "s_mov_b64 exec, s[92:93]\n\t", 256)

// taken branches: K e32 adds (4 B each), then an unconditional branch to the next group; target aligned to 64 B,
// or deliberately 4 B before a 64 B boundary
#define GROUP_ALIGNED(K) ".rept 16\n\t.p2align 6\n\t.rept " #K "\n\tv_add_u32_e32 %0, %13, %0\n\t.endr\n\ts_branch 1f\n\t.p2align 6\n\t1:\n\t.endr\n\t"
#define GROUP_UNALIGNED(K) ".rept 16\n\t.rept " #K "\n\tv_add_u32_e32 %0, %13, %0\n\t.endr\n\ts_branch 1f\n\t.p2align 6\n\t.rept 15\n\ts_nop 0\n\t.endr\n\t1:\n\t.endr\n\t"
TIMED(k_br_a8, GROUP_ALIGNED(8), 16 * 9)
TIMED(k_br_a32, GROUP_ALIGNED(32), 16 * 33)
TIMED(k_br_u8, GROUP_UNALIGNED(8), 16 * 9)
TIMED(k_br_u32, GROUP_UNALIGNED(32), 16 * 33)
// not-taken conditional branches between groups (what the unrolled squaring run pays)
#define GROUP_NT(K) ".rept 16\n\t.rept " #K "\n\tv_add_u32_e32 %0, %13, %0\n\t.endr\n\ts_cmp_eq_u32 s90, 0\n\ts_cbranch_scc1 8f\n\t.endr\n\t8:\n\t"
TIMED(k_nt8, GROUP_NT(8), 16 * 10)

// loop-head alignment: the loop of TIMED sits at a 64 B boundary; here an inner loop head at byte offset 4 K of a line,
// its body 96 eight-byte instructions (like a squaring of the fold product) closed by a taken branch
#define HEAD(K, NAME)                                                                                       \
  TIMED(NAME, "s_mov_b32 s91, 8\n\t.p2align 6\n\t.rept " #K "\n\ts_nop 0\n\t.endr\n\t5:\n\t"                    \
              ".rept 12\n\t" R8(DPP) ".endr\n\t"                                                            \
              "s_sub_u32 s91, s91, 1\n\ts_cmp_lg_u32 s91, 0\n\ts_cbranch_scc1 5b\n\t", 8 * 99 + K)
HEAD(0, k_h0) HEAD(1, k_h1) HEAD(2, k_h2) HEAD(3, k_h3) HEAD(4, k_h4) HEAD(5, k_h5) HEAD(6, k_h6) HEAD(7, k_h7)
HEAD(8, k_h8) HEAD(9, k_h9) HEAD(10, k_h10) HEAD(11, k_h11) HEAD(12, k_h12) HEAD(13, k_h13) HEAD(14, k_h14) HEAD(15, k_h15)

struct Case { const char* name; void (*fn)(unsigned long long*, uint32_t, int); };
int main() {
  Case cases[] = {{"v_add_u32_e32 (4 B)", k_add32}, {"v_add_u32_e64 (8 B)", k_add64}, {"v_and_b32 literal (8 B)", k_andlit},
                  {"v_and_b32 SGPR (4 B)", k_andsgpr}, {"v_mov_b32_dpp (8 B)", k_dpp}, {"v_add_u32_dpp (8 B)", k_adddpp},
                  {"s_nop 0 (4 B)", k_nop}, {"s_add_u32 (4 B)", k_salu}, {"v_mad_u64_u32 (8 B)", k_mad},
                  {"mad, dpp, add32 mix", k_mix}, {"mad, add32, add32 mix", k_mix32},
                  {"v_permlane16_swap_b32", k_swap16}, {"v_permlane32_swap_b32", k_swap32}, {"v_mov_b32_dpp row_newbcast", k_bcast},
                  {"v_mov_b32_dpp row_ror bank-masked", k_ror}, {"v_lshl_add_u64", k_lshladd64}, {"v_alignbit_b32", k_alignbit},
                  {"v_mad_u64_u32, ONE dependent chain", k_mad_dep}, {"v_add_u32, ONE dependent chain", k_add_dep},
                  {"mad, add, add, dpp(of older reg)", k_mad_then_dpp}, {"add a, add, add, dpp reads a", k_add_2_dpp},
                  {"add a, s_nop 1, dpp reads a, add", k_add_nop_dpp}, {"add a, add, add, swap16 a", k_add_2_swap},
                  {"mad, 64-bit add of its result", k_mad_use},
                  {"v_nop_e64 (8 B)", k_vnop64}, {"v_nop (4 B)", k_vnop32}, {"v_sub_u32_e64 (8 B)", k_sub64},
                  {"add a, mad, s_nop 0 x2, dpp reads a, 3 mads", k_gap_snop}, {"add a, mad, v_nop_e64, dpp reads a, 3 mads", k_gap_vnop},
                  {"(v_add, s_add, s_nop 0) x", k_vs_sadd}, {"(v_add, s_mov literal) x", k_vs_smovlit},
                  {"(v_add, s_nop 0, s_nop 0) x", k_vs_snop}, {"(v_add, s_cmp, s_nop 0) x", k_vs_scmp},
                  {"add a, mad, s_mov literal, dpp reads a, 3 mads", k_gap_smov}, {"add a, mad, s_nop 0, dpp reads a, 3 mads", k_gap_snop0},
                  {"v_lshrrev_b64 (8 B)", k_lshr64}, {"v_mul_lo_u32 (8 B)", k_mullo}, {"7 dependent mads + v_lshrrev_b64", k_column},
                  {"v_add_u32, EXEC = low 32 lanes", k_exec32}, {"v_add_u32, EXEC = low 16 lanes", k_exec16},
                  {"v_mad_u64_u32, EXEC = low 32 lanes", k_exec32_mad}, {"v_mov_b32_dpp, EXEC = low 32 lanes", k_exec32_dpp},
                  {"8 adds + taken branch, aligned", k_br_a8}, {"32 adds + taken branch, aligned", k_br_a32},
                  {"8 adds + taken branch, target at line end", k_br_u8}, {"32 adds + taken, target at line end", k_br_u32},
                  {"8 adds + cmp + not-taken branch", k_nt8},
                  {"96 x 8 B + taken branch, head at +0", k_h0}, {"  head at +4", k_h1}, {"  head at +8", k_h2}, {"  head at +12", k_h3},
                  {"  head at +16", k_h4}, {"  head at +20", k_h5}, {"  head at +24", k_h6}, {"  head at +28", k_h7},
                  {"  head at +32", k_h8}, {"  head at +36", k_h9}, {"  head at +40", k_h10}, {"  head at +44", k_h11},
                  {"  head at +48", k_h12}, {"  head at +52", k_h13}, {"  head at +56", k_h14}, {"  head at +60", k_h15}};
  unsigned long long* d;
  (void)hipMalloc(&d, 64);
  printf("%-46s %10s %12s %8s\n", "one wavefront, one workgroup", "cycles/instr", "cycles/iter", "GHz");
  for (auto& c : cases) {
    unsigned long long h[3];
    std::vector<double> v;
    double ghz = 0, per = 0;
    for (int rep = 0; rep < 5; rep++) {
      hipLaunchKernelGGL(c.fn, dim3(1), dim3(64), 0, 0, d, 12345u, 2000);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      v.push_back((double)h[0] / 2000.0);
      ghz = (double)h[0] / (double)h[1] * 0.1;
      per = (double)h[2];
    }
    std::sort(v.begin(), v.end());
    printf("%-46s %10.2f %12.1f %8.3f\n", c.name, v[2] / (per + 3), v[2], ghz);
  }
  printf("status: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
