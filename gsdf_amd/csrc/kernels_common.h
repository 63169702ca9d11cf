// kernels_common.h -- what every kernel file of the MI355X SDF backend shares: the workgroup size, the dynamic LDS block, the
// octree cube and the mesher's device counters, wave-level append, the multi-GPU brick owner. Device code only (compiled ahead
// of time by hipcc and at run time for one lowered program, see kernels.h).
#pragma once
#include "interp.h"
#include "mc_tables.h"

using gsdf_dev::code_ptr;
using gsdf_dev::P3;

#define BLOCK 256
#define LEAF_MIN_COLS 14  // leaf_kernel's LDS columns per lane: 8 corner distances + 3 origin + 3 for the owner list / cube indices
#define TRI_STAGE 128  // triangles staged in LDS per workgroup before one coalesced flush (4.5 KB: lets 4 workgroups of a 7-slot program share a CU)

// ---------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ code_ptr as_code(const uint32_t* p) { return (code_ptr)(uintptr_t)p; }

extern __shared__ __attribute__((aligned(16))) float g_smem[];

// Octree cube: level-index coordinates (leaf coordinate >> (level-1)).
struct __attribute__((aligned(8))) Cube {
  uint16_t x, y, z, w;
};

#define MAX_LEVELS 24
struct MeshCounters {
  unsigned long long n_level[MAX_LEVELS];  // [L]: cubes of level L handed to the next stage (survivors kept by this rank)
  unsigned long long n_items[MAX_LEVELS];  // [L]: candidate cubes centre-tested at level L (0 if the level was not tested)
  unsigned long long n_pass[MAX_LEVELS];   // [L]: candidates that passed the prune predicate (before shard filter)
  unsigned long long n_active;             // leaves passing the corner-0 test
  unsigned long long pad0[16];             // the triangle append counter gets a cache line (L2 atomic unit) of its own
  unsigned long long n_tris;
  unsigned long long pad1[15];
  unsigned long long overflow;             // triangle buffer overflow flag
  unsigned long long n_cont;               // leaves whose wave went on to the remaining corners
  unsigned long long q_overflow;           // cube queue capacity exceeded
  unsigned long long n_points;             // lattice points evaluated by leaf_brick_kernel
  unsigned long long n_cut;                // leaves the surface cuts (records written by leaf_eval_kernel)
};

// Decisions on a value that may be NaN (a distance of a degenerate tree, tests/test_gpu_nan.py), as CLASS / INTEGER tests on its
// bits. The specialised kernels are built with -fno-honor-nans (specialize.cpp), under which a float comparison with a NaN
// operand is whatever is cheapest: `!(|d| >= m)` came out as `|d| < m` and dropped the cubes the reference keeps, `v != v` was
// folded away. These say what IEEE comparisons say (false for NaN) in every build, at the same instruction count
// (v_cmp_class_f32; one v_and + v_cmp_u32 for the magnitude tests).
namespace nb {
// v_cmp_class_f32 mask bits: 0 sNaN, 1 qNaN, 2 -inf, 3 -normal, 4 -subnormal, 5 -0, 6 +0, 7 +subnormal, 8 +normal, 9 +inf
__device__ __forceinline__ bool lt0(float v) { return __builtin_amdgcn_classf(v, 0x01c); }   // v <  0
__device__ __forceinline__ bool le0(float v) { return __builtin_amdgcn_classf(v, 0x07c); }   // v <= 0
__device__ __forceinline__ bool ge0(float v) { return __builtin_amdgcn_classf(v, 0x3e0); }   // v >= 0
__device__ __forceinline__ bool nan_or_inf(float v) { return __builtin_amdgcn_classf(v, 0x207); }
// |v| <= lim and |v| >= lim for a finite lim >= 0: non-negative floats order like their bit patterns
__device__ __forceinline__ bool abs_le(float v, float lim) { return (__float_as_uint(v) & 0x7fffffffu) <= __float_as_uint(lim); }
__device__ __forceinline__ bool abs_ge(float v, float lim) {
  const unsigned a = __float_as_uint(v) & 0x7fffffffu;
  return a >= __float_as_uint(lim) && a <= 0x7f800000u;
}
}  // namespace nb

// wave64 compaction: returns the global slot for lanes with keep=true (others undefined).
__device__ __forceinline__ unsigned long long wave_append(bool keep, unsigned long long* counter) {
  const unsigned long long mask = __ballot(keep);
  const unsigned int lane_prefix = __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
  unsigned long long base = 0;
  if (mask != 0ull) {
    const int leader = __builtin_ctzll(mask);
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(counter, (unsigned long long)__builtin_popcountll(mask));
    base = __shfl(base, leader, 64);
  }
  return base + lane_prefix;
}

// Block-wide compaction of a variable number of items per thread (0 .. a few): returns the global slot of the calling
// thread's first item; its `mine` items take consecutive slots. ONE returning atomic per call and workgroup -- a single counter
// word serves ~88 returning atomics per microsecond on MI355X, so per-wave appends bound kernels that append from every wave
// pass (dual contouring's edge and quad stages: 48 K / 16 K of them per mesh, 0.55 / 0.18 ms of atomics alone). Every thread
// of the workgroup must call it the same number of times (it synchronises); `total` (optional) = the workgroup's item count.
__device__ __forceinline__ unsigned long long block_append_n(unsigned mine, unsigned long long* counter, unsigned* total_out = nullptr) {
  __shared__ unsigned s_w[4];
  __shared__ unsigned long long s_base;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  unsigned incl = mine;  // wave inclusive scan of the per-lane counts
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned v = __shfl_up(incl, off, 64);
    if (lane >= (unsigned)off) incl += v;
  }
  __syncthreads();  // the previous call's readers are done with s_w / s_base
  if (lane == 63u) s_w[wave] = incl;
  __syncthreads();
  const unsigned w0 = s_w[0], w1 = s_w[1], w2 = s_w[2], w3 = s_w[3];
  const unsigned total = w0 + w1 + w2 + w3;
  if (threadIdx.x == 0 && total) s_base = atomicAdd(counter, (unsigned long long)total);
  __syncthreads();
  if (total_out) *total_out = total;
  return (total ? s_base : 0ull) + (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u) + (incl - mine);
}

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// Owner rank of a brick for multi-GPU sharding: a pure function of the brick coordinates, so every
// rank derives the same partition with no communication and no ordering dependence.
__host__ __device__ __forceinline__ unsigned brick_owner(unsigned x, unsigned y, unsigned z, unsigned count) {
  unsigned h = (x * 73856093u) ^ (y * 19349663u) ^ (z * 83492791u);
  h ^= h >> 15;
  h *= 0x2c1b3c6du;
  h ^= h >> 12;
  return h % count;
}
