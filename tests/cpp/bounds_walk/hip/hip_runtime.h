// Stand-in for <hip/hip_runtime.h> when the KERNEL BODIES of anemoi-rust_amd/csrc are compiled for the host by
// tests/cpp/bounds_walk/bounds_walk.cpp (g++, -DANEMOI_BOUNDS_WALK): just enough of the HIP device vocabulary to run
// a kernel function as ordinary C++, one coroutine per emulated lane.  Test infrastructure; never part of the product.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__

struct uint4 {
  uint32_t x, y, z, w;
};
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

struct WalkDim3 {
  unsigned x, y, z;
};
extern WalkDim3 threadIdx, blockIdx, gridDim;   // set by the lane scheduler at every switch

// cross-lane primitives: implemented by the walker (a rendezvous of the emulated lanes)
namespace walk {
uint32_t exchange(uint32_t mine, int src_lane);   // every emulated lane calls it; returns src_lane's `mine` (own if absent)
void barrier();
}  // namespace walk

static inline void __syncthreads() { walk::barrier(); }
static inline int __shfl(int v, int src_lane) { return (int)walk::exchange((uint32_t)v, src_lane); }
static inline int __shfl_xor(int v, int mask) { return (int)walk::exchange((uint32_t)v, int(threadIdx.x) ^ mask); }
static inline unsigned __shfl(unsigned v, int src_lane) { return walk::exchange(v, src_lane); }
static inline unsigned __shfl_xor(unsigned v, int mask) { return walk::exchange(v, int(threadIdx.x) ^ mask); }
static inline unsigned long __shfl_xor(unsigned long v, int mask) {   // (64-bit block counts of the ragged sponge: two halves)
  const unsigned long lo = walk::exchange((uint32_t)v, int(threadIdx.x) ^ mask);
  const unsigned long hi = walk::exchange((uint32_t)(v >> 32), int(threadIdx.x) ^ mask);
  return hi << 32 | lo;
}
// DPP: only the control the lane-private kernels use (quad_perm:[1,0,3,2] = swap with the pair neighbour)
static inline int __builtin_amdgcn_update_dpp(int, int src, int ctrl, int, int, bool) {
  if (ctrl == 0xB1) return (int)walk::exchange((uint32_t)src, int(threadIdx.x) ^ 1);
  __builtin_trap();
}
static inline uint32_t __builtin_amdgcn_s_getreg(int) { return 0; }
static inline void __builtin_amdgcn_s_setprio(int) {}
