// kernels.h -- every gfx950 kernel of the MI355X SDF backend (device code only).
//
// Compiled twice: ahead of time by hipcc as part of gsdf_hip.hip (evaluator = the wave-uniform interpreter of
// interp.h), and at run time by hiprtc for one lowered program (GSDF_SPECIALIZED: interp.h then takes sdf_eval from
// the generated gsdf_spec_gen.h, the same instruction bodies laid out straight-line with every parameter a literal).
// Nothing here may depend on host headers.
//
// Kernels (all wave64, 256-thread workgroups, grid-stride with wave-uniform trip counts so that the
// interpreter's program counter stays scalar):
//   eval_kernel<DIM>     dist[i] = SDF(pos[i])                      gleval SDF3/SDF2.Evaluate
//   prune_kernel         octree level: centre sample, keep unless the field's bounds over the cube exclude 0
//                        (|d| >= size*sqrt3/2 for a distance field), block-wide compaction of survivors
//                                                                   glrender/octreerenderer.go:240-284
//   leaf_kernel          8 leaf corners (corner 0 first, reject |d0| > 2*sqrt3*res) + marching cubes
//                        with the LDS triangle table; triangles are built one per lane from an LDS
//                        owner list (mc_emit_balanced), staged in LDS and flushed coalesced
//                                                                   glrender/marchcubes.go:14-98
//   flat_grid_kernel     SDF on every corner of the flat lattice    glrender/flatrenderer.go:103-182
//                        (+ two bits per corner: d < 0, |d| <= 2 sqrt3 res)
//   flat_cut_scan_kernel, flat_march_list_kernel
//                        marching cubes of every lattice cube: cut cubes found from the bit planes, listed, marched
//                                                                   glrender/flatrenderer.go:186-256
//   flat_march_kernel    the same from the float grid alone (GSDF_HIP_FLAT_STREAM=1; rounds 1-2)
//   dc_*_kernel          dual contouring stages                     glrender/dual_contour*.go
//   stl_kernel           50-byte STL records staged through LDS     glrender/stl.go:15-62
//   normals_kernel       central differences                        gleval/gleval.go:53-108
#pragma once
#include "interp.h"
#include "mc_tables.h"

using gsdf_dev::code_ptr;
using gsdf_dev::P3;

#define BLOCK 256

// ---------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ code_ptr as_code(const uint32_t* p) { return (code_ptr)(uintptr_t)p; }

extern __shared__ __attribute__((aligned(16))) float g_smem[];

// dist[i] = SDF(pos[i]). Each lane carries K points per interpreter pass (tile = K*BLOCK points,
// point kp of lane t = tile + kp*BLOCK + t, so every load/store stays coalesced).
// W = workgroups per CU the register budget is sized for: the host asks for 4 (<= 128 VGPRs) whenever 4 workgroups'
// slot columns fit the 160 KB of LDS -- the 4th wave per SIMD is worth more than the spills (interpreter build of
// npt-flange: flat lattice 63 -> 73 G evals/s).
template <int DIM, int K, int W = (K == 1 ? 4 : 3)>
__global__ void __launch_bounds__(BLOCK, W) eval_kernel(const uint32_t* __restrict__ code_g, const float* __restrict__ pos,
                                                     uint32_t stride_f, float* __restrict__ dist, uint64_t n) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * BLOCK * K;
  for (uint64_t base = (uint64_t)blockIdx.x * BLOCK * K; base < n; base += step) {  // uniform trip count
    P3 p[K];
    float d[K];
#pragma unroll
    for (int kp = 0; kp < K; kp++) {
      const uint64_t i = base + (uint64_t)kp * BLOCK + threadIdx.x;
      p[kp] = P3{0.f, 0.f, 0.f};
      if (i < n) {
        const float* q = pos + i * stride_f;
        p[kp].x = q[0];
        p[kp].y = q[1];
        if (DIM == 3) p[kp].z = q[2];
      }
    }
    gsdf_dev::sdf_eval<K>(code, p, d, lds, BLOCK);
#pragma unroll
    for (int kp = 0; kp < K; kp++) {
      const uint64_t i = base + (uint64_t)kp * BLOCK + threadIdx.x;
      if (i < n) dist[i] = d[kp];
    }
  }
}

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

// One octree level, chained on the stream with NO host round trip: the candidate count is read from the
// previous level's survivor counter in device memory. expand=1: item i is child (i&7) of in[i>>3];
// expand=0: the single top cube. Survivors are compacted block-wide (ballot + mbcnt prefix per wave, 4 wave totals
// through LDS) into an LDS stage of PRUNE_STAGE cubes and appended to `out` with ONE global atomic per flush: a
// single counter word takes ~88 atomics/us on MI355X, so the per-wave appends of the first version bounded the two
// big levels (8940 waves at level 3 = 100 us of a 124 us kernel).
// The test (do_test = 1): the field's bounds over the cube's bounding ball, by interval evaluation at the centre (interp.h: LIP;
// two "points" per lane), exclude 0 -- for a true distance field exactly the reference's |d| >= size * sqrt3/2, and still
// surface-preserving for twists, screws and non-rigid transforms (dev_ops.h: D_LIP_*). do_test = 2: the reference's predicate
// verbatim on the centre value, whatever the field (gsdf_mesh_opts.prune: GSDF_PRUNE_ASSUME_SDF).
// LDS: [2 * ncols floats per lane (ncols = program slots + interval stack) | PRUNE_STAGE cubes | 4 wave totals | base].
#define PRUNE_STAGE 1024
__global__ void __launch_bounds__(BLOCK) prune_kernel(const uint32_t* __restrict__ code_g, const Cube* __restrict__ in,
                                                      unsigned long long in_cap, int expand, int level, int ncols, int lip_base, float ox,
                                                      float oy, float oz, float res,
                                                      int do_test, Cube* __restrict__ out, unsigned long long out_cap,
                                                      int shard_here, unsigned shard_rank, unsigned shard_count,
                                                      MeshCounters* __restrict__ ctr) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  Cube* s_q = (Cube*)(g_smem + (size_t)(ncols > 0 ? ncols : 1) * 2 * BLOCK);
  unsigned* s_w = (unsigned*)(s_q + PRUNE_STAGE);  // [0..3] wave totals, [4..7] per-wave "passed the test" counts
  unsigned long long* s_base = (unsigned long long*)(s_w + 8);
  // the previous level counts every survivor, also those its queue had no room for (the host then grows the queues
  // and reruns): never read past what was stored
  unsigned long long n_in = expand ? uniform_u64(ctr->n_level[level + 1]) : 0ull;
  if (n_in > in_cap) n_in = in_cap;
  const unsigned long long n_items = expand ? n_in * 8ull : 1ull;
  if (blockIdx.x == 0 && threadIdx.x == 0) ctr->n_items[level] = do_test ? n_items : 0ull;
  const float size = (float)(1 << (level - 1)) * res;  // i3.Cube size at this level
  const float maxDist = size * (1.73205080757f / 2);    // szDistMult = sqrt3/2 (octreerenderer.go:182)
  const uint64_t step = (uint64_t)gridDim.x * BLOCK;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long my_pass = 0;
  unsigned cur = 0;  // cubes staged so far (block-uniform: every thread derives it from the same LDS totals)
  auto flush = [&]() {  // block-uniform
    if (threadIdx.x == 0) *s_base = atomicAdd(&ctr->n_level[level], (unsigned long long)cur);
    __syncthreads();
    const unsigned long long fb = *s_base;
    if (fb + cur <= out_cap) {
      for (unsigned k = threadIdx.x; k < cur; k += BLOCK) out[fb + k] = s_q[k];
    } else if (threadIdx.x == 0) {
      ctr->q_overflow = 1ull;
    }
    __syncthreads();
    cur = 0;
  };
  for (uint64_t base = (uint64_t)blockIdx.x * BLOCK; base < n_items; base += step) {  // block-uniform trip count
    const uint64_t i = base + threadIdx.x;
    const bool valid = i < n_items;
    Cube c = {0, 0, 0, 0};
    if (valid && expand) {
      const Cube pc = in[i >> 3];
      const unsigned k = (unsigned)(i & 7);
      // children in corner order: 0:(0,0,0) 1:(+x) 2:(+x,+y) 3:(+y) 4..7 same at +z
      c.x = (uint16_t)(pc.x * 2 + ((k ^ (k >> 1)) & 1));
      c.y = (uint16_t)(pc.y * 2 + ((k >> 1) & 1));
      c.z = (uint16_t)(pc.z * 2 + ((k >> 2) & 1));
    }
    bool keep = valid;
    if (do_test) {
      const float cx0 = ox + size * (float)c.x, cy0 = oy + size * (float)c.y, cz0 = oz + size * (float)c.z;
      P3 p;  // CubeCenter = Scale(0.5, Add(min, max)), max = min + size
      p.x = 0.5f * (cx0 + (cx0 + size));
      p.y = 0.5f * (cy0 + (cy0 + size));
      p.z = 0.5f * (cz0 + (cz0 + size));
      P3 pv[2] = {p, p};
      float dv[2];
      if (do_test == 2) {
        gsdf_dev::sdf_eval<2>(code, pv, dv, lds, BLOCK);
        keep = valid && !(dm::absf(dv[0]) >= maxDist);
      } else {
        gsdf_dev::sdf_eval<2, 0, true>(code, pv, dv, lds, BLOCK, false, maxDist, (uint32_t)lip_base);
        keep = valid && !(dv[0] >= 0.0f || dv[1] <= 0.0f);
      }
    }
    const unsigned long long pm = __ballot(keep);
    if (lane == 0) my_pass += (unsigned long long)__builtin_popcountll(pm);
    if (shard_here) keep = keep && (brick_owner(c.x, c.y, c.z, shard_count) == shard_rank);
    const unsigned long long km = __ballot(keep);
    const unsigned lane_prefix = __builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u));
    if (lane == 0) s_w[wave] = (unsigned)__builtin_popcountll(km);
    __syncthreads();
    const unsigned w0 = s_w[0], w1 = s_w[1], w2 = s_w[2], w3 = s_w[3];
    const unsigned total = w0 + w1 + w2 + w3;
    const unsigned wpre = (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u);
    if (cur + total > PRUNE_STAGE) flush();  // total <= 256 always fits afterwards
    if (keep) s_q[cur + wpre + lane_prefix] = c;
    cur += total;
    __syncthreads();  // s_w is rewritten next iteration; s_q complete before a flush reads it
  }
  if (cur) flush();
  // statistics: one atomic per workgroup (the kernel cannot retire before its atomics do: 4 per workgroup on one
  // word were 46 us of the level-3 launch)
  if (lane == 0) s_w[4 + wave] = (unsigned)my_pass;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = (unsigned long long)s_w[4] + s_w[5] + s_w[6] + s_w[7];
    if (t) atomicAdd(&ctr->n_pass[level], t);
  }
}

// The top of the octree WITHOUT its chain of dependent launches. A level is centre-tested after its parent level because a
// dropped cube's children need no test -- an economy that is worth nothing where a level is a few thousand cubes and its launch
// 10 us of latency (npt-flange at resdiv 1600: 10 levels, 0.1 ms of a 0.7 ms mesh). The first S levels (all the cubes of Levels
// top .. top-S+1: (8^S - 1) / 7 of them, 299,593 for S = 7 -- 0.2 % of that mesh's evaluations) are therefore tested
// SPECULATIVELY, every cube of the complete octree at once, in one launch (prune_spec_kernel: a byte per cube), and a second
// launch keeps the cubes whose ancestors all passed (prune_resolve_kernel: ancestor bytes are independent loads, no
// level-by-level pass) and compacts the survivors of the last speculative level into the queue the per-level kernels continue
// from. Same tests on the same centres, so the survivors are exactly the per-level chain's; the counters count the cubes that
// chain would have tested (candidates: children of survivors), not the speculative ones.
// Cube number i of the speculative block: level top-j for off(j) <= i < off(j+1), off(j) = (8^j - 1) / 7; k = i - off(j) spells
// the path from the top cube in base 8, most significant digit first, a digit being the child number in corner order; the
// parent of (j, k) is (j-1, k >> 3).
__device__ __forceinline__ unsigned spec_level_of(unsigned i, unsigned& k) {  // j and the number within the level
  unsigned j = 0, off = 0, n = 1;
  while (i >= off + n) { off += n; n <<= 3; j++; }  // <= 7 steps
  k = i - off;
  return j;
}
__device__ __forceinline__ Cube spec_cube(unsigned j, unsigned k) {
  unsigned x = 0, y = 0, z = 0;
  for (unsigned d = 0; d < j; d++) {
    const unsigned c = (k >> (3u * (j - 1u - d))) & 7u;
    x = 2u * x + ((c ^ (c >> 1)) & 1u);
    y = 2u * y + ((c >> 1) & 1u);
    z = 2u * z + ((c >> 2) & 1u);
  }
  Cube c = {(uint16_t)x, (uint16_t)y, (uint16_t)z, 0};
  return c;
}
// pass[i]: bit 0 = the cube passed its centre test (or its level is not tested), bit 1 = this rank owns it (multi-GPU: at the
// level where bricks are dealt to ranks; everywhere else set).
// LDS: [2 * ncols floats per lane] (interval mode, see prune_kernel).
__global__ void __launch_bounds__(BLOCK) prune_spec_kernel(const uint32_t* __restrict__ code_g, int top, unsigned n_spec, int ncols,
                                                           int lip_base, float ox, float oy, float oz, float res, unsigned test_mask,
                                                           int ptest, int shard_level, unsigned shard_rank, unsigned shard_count,
                                                           uint8_t* __restrict__ pass) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  (void)ncols;
  const unsigned step = gridDim.x * BLOCK;
  for (unsigned base = blockIdx.x * BLOCK; base < n_spec; base += step) {  // block-uniform trip count
    const unsigned i = base + threadIdx.x;
    const bool valid = i < n_spec;
    unsigned k = 0;
    const unsigned j = spec_level_of(valid ? i : 0u, k);
    const int level = top - (int)j;
    const Cube c = spec_cube(j, k);
    const bool tested = level >= 3 && ((test_mask >> level) & 1u) != 0u;
    const float size = (float)(1 << (level - 1)) * res;
    const float maxDist = size * (1.73205080757f / 2);
    const float cx0 = ox + size * (float)c.x, cy0 = oy + size * (float)c.y, cz0 = oz + size * (float)c.z;
    P3 p;
    p.x = 0.5f * (cx0 + (cx0 + size));
    p.y = 0.5f * (cy0 + (cy0 + size));
    p.z = 0.5f * (cz0 + (cz0 + size));
    P3 pv[2] = {p, p};
    float dv[2];
    bool keep;
    if (ptest == 2) {
      gsdf_dev::sdf_eval<2>(code, pv, dv, lds, BLOCK);
      keep = !(dm::absf(dv[0]) >= maxDist);
    } else {
      gsdf_dev::sdf_eval<2, 0, true>(code, pv, dv, lds, BLOCK, false, maxDist, (uint32_t)lip_base);  // (maxDist differs from lane to lane: fine, it is the lane's own radius)
      keep = !(dv[0] >= 0.0f || dv[1] <= 0.0f);
    }
    if (!tested) keep = true;
    const bool own = level != shard_level || brick_owner(c.x, c.y, c.z, shard_count) == shard_rank;
    if (valid) pass[i] = (uint8_t)((keep ? 1u : 0u) | (own ? 2u : 0u));
  }
}

// Survivors of the speculative block: cube i lives on iff it and every ancestor passed (and was owned). Every thread walks its
// own ancestor chain -- up to S independent byte loads from a table that sits in L2 -- so there is no pass per level. The
// survivors of the LAST speculative level are staged in LDS and appended to `out` with one atomic per workgroup; per level,
// the candidates the per-level chain would have tested (children of survivors; the top cube itself) and those that passed are
// added to the counters with one atomic per workgroup and level that saw any.
#define SPEC_STAGE 2048
__global__ void __launch_bounds__(BLOCK) prune_resolve_kernel(const uint8_t* __restrict__ pass, int top, int S, unsigned n_spec,
                                                              unsigned test_mask, Cube* __restrict__ out, unsigned long long out_cap,
                                                              MeshCounters* __restrict__ ctr) {
  __shared__ Cube s_q[SPEC_STAGE];
  __shared__ unsigned s_n, s_items[8], s_pass[8];
  __shared__ unsigned long long s_base;
  if (threadIdx.x == 0) s_n = 0;
  if (threadIdx.x < 8) { s_items[threadIdx.x] = 0; s_pass[threadIdx.x] = 0; }
  __syncthreads();
  const unsigned per = (n_spec + gridDim.x - 1) / gridDim.x;  // a contiguous slice per workgroup
  const unsigned i0 = blockIdx.x * per, i1 = i0 + per < n_spec ? i0 + per : n_spec;
  for (unsigned base = i0; base < i1; base += BLOCK) {
    const unsigned i = base + threadIdx.x;
    if (i >= i1) continue;
    unsigned k = 0;
    const unsigned j = spec_level_of(i, k);
    // ancestors: (j-1, k>>3), (j-2, k>>6) ... ; off(j) = (8^j - 1) / 7
    unsigned anc = 1u;  // every proper ancestor passed and was owned (no short circuit: the loads are independent)
    unsigned kk = k, off = i - k;
    for (unsigned a = j; a > 0; a--) {
      kk >>= 3;
      off = (off - 1u) >> 3;  // off(a-1) = (off(a) - 1) / 8
      anc &= pass[off + kk] == 3u ? 1u : 0u;
    }
    const unsigned me = pass[i];
    if (anc) {  // a candidate of the per-level chain
      atomicAdd(&s_items[j], 1u);
      if (me & 1u) atomicAdd(&s_pass[j], 1u);
      if (me == 3u && (int)j == S - 1) {
        const unsigned slot = atomicAdd(&s_n, 1u);
        if (slot < SPEC_STAGE) s_q[slot] = spec_cube(j, k);
      }
    }
  }
  __syncthreads();
  const unsigned n = s_n;
  if (threadIdx.x == 0) {
    s_base = n ? atomicAdd(&ctr->n_level[top - (S - 1)], (unsigned long long)n) : 0ull;
    for (int j = 0; j < S && j < 8; j++) {
      const int level = top - j;
      const bool tested = level >= 3 && ((test_mask >> level) & 1u) != 0u;
      if (s_items[j] && tested) atomicAdd(&ctr->n_items[level], (unsigned long long)s_items[j]);
      if (s_pass[j]) atomicAdd(&ctr->n_pass[level], (unsigned long long)s_pass[j]);
    }
    if (n > SPEC_STAGE) ctr->q_overflow = 1ull;  // cannot happen: the host sizes the grid so that a slice is at most SPEC_STAGE cubes
  }
  __syncthreads();
  const unsigned long long fb = s_base;
  if (fb + n <= out_cap) {
    for (unsigned q = threadIdx.x; q < n && q < SPEC_STAGE; q += BLOCK) out[fb + q] = s_q[q];
  } else if (threadIdx.x == 0) {
    ctr->q_overflow = 1ull;
  }
}

// mcInterpolate (marchcubes.go:76-98) with x = 0.
__device__ __forceinline__ void mc_interp(float ax, float ay, float az, float bx, float by, float bz, float v1, float v2,
                                          float& rx, float& ry, float& rz) {
  const float eps = 1e-12f;
  const bool c1 = dm::absf(0.f - v1) < eps, c2 = dm::absf(0.f - v2) < eps;
  float t = 0.5f;
  if (!c1 || !c2) t = (0.f - v1) / (v2 - v1);
  float x = ax + t * (bx - ax), y = ay + t * (by - ay), z = az + t * (bz - az);
  if (c1 && !c2) { x = ax; y = ay; z = az; }
  if (c2 && !c1) { x = bx; y = by; z = bz; }
  rx = x; ry = y; rz = z;
}

#define LEAF_MIN_COLS 14  // leaf_kernel's LDS columns per lane: 8 corner distances + 3 origin + 3 for the owner list / cube indices
#define TRI_STAGE 128  // triangles staged in LDS per workgroup before one coalesced flush (4.5 KB: lets 4 workgroups of a 7-slot program share a CU)

// Marching cubes of one leaf per lane + block-wide triangle emission (shared by both leaf kernels).
// vslot: the lane's 8 corner distances in its LDS column; index: the 8-bit inside mask (0 = no triangles).
// Block-uniform control flow: every thread of the workgroup must call this the same number of times.
template <int STAGE = TRI_STAGE, typename CornerDist>
__device__ __forceinline__ void mc_emit_block(unsigned index, float x0, float y0, float z0, float x1, float y1, float z1,
                                              CornerDist vdist, const int8_t* s_tri, float* s_stage, unsigned* s_misc,
                                              unsigned long long* s_base, float* __restrict__ tris, uint64_t tri_cap,
                                              MeshCounters* __restrict__ ctr) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned nt = 0;
  {
    const int8_t* row = s_tri + index * 16;
    while (nt < 5 && row[3 * nt] >= 0) nt++;
  }
  // block exclusive scan of nt: wave scan + 4 wave totals through LDS
  unsigned incl = nt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    unsigned v = __shfl_up(incl, off, 64);
    if (lane >= (unsigned)off) incl += v;
  }
  if (lane == 63) s_misc[wave] = incl;
  __syncthreads();  // (A)
  const unsigned w0 = s_misc[0], w1 = s_misc[1], w2 = s_misc[2], w3 = s_misc[3];
  const unsigned total = w0 + w1 + w2 + w3;
  const unsigned wpre = (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u);
  unsigned cur = s_misc[4];
  const bool direct = total > STAGE;  // block-uniform
  unsigned long long gbase = 0;
  if (!direct && cur + total > STAGE) {  // flush the stage first (block-uniform)
    if (threadIdx.x == 0) *s_base = atomicAdd(&ctr->n_tris, (unsigned long long)cur);
    __syncthreads();
    const unsigned long long fb = *s_base;
    if (fb + cur <= tri_cap) {
      float* dst = tris + fb * 9;
      for (unsigned k = threadIdx.x; k < cur * 9; k += BLOCK) dst[k] = s_stage[k];
    } else if (threadIdx.x == 0) {
      ctr->overflow = 1ull;
    }
    __syncthreads();
    cur = 0;
  }
  if (direct) {
    if (threadIdx.x == 0) *s_base = atomicAdd(&ctr->n_tris, (unsigned long long)total);
    __syncthreads();
    gbase = *s_base;
    if (gbase + total > tri_cap) {
      if (threadIdx.x == 0) ctr->overflow = 1ull;
      nt = 0;
    }
  }
  if (nt) {
    const unsigned first = wpre + (incl - nt);
    float* dst = direct ? (tris + (gbase + first) * 9) : (s_stage + (size_t)(cur + first) * 9);
    const int8_t* row = s_tri + index * 16;
    for (unsigned t = 0; t < nt; t++) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int e = row[3 * t + (2 - k)];  // reversed winding (marchcubes.go:64-68)
        const unsigned a = GSDF_MC_PAIR_A(e), b = GSDF_MC_PAIR_B(e);
        const float va = vdist(a), vb = vdist(b);
        const float pax = ((a ^ (a >> 1)) & 1u) ? x1 : x0, pay = ((a >> 1) & 1u) ? y1 : y0, paz = ((a >> 2) & 1u) ? z1 : z0;
        const float pbx = ((b ^ (b >> 1)) & 1u) ? x1 : x0, pby = ((b >> 1) & 1u) ? y1 : y0, pbz = ((b >> 2) & 1u) ? z1 : z0;
        float rx, ry, rz;
        mc_interp(pax, pay, paz, pbx, pby, pbz, va, vb, rx, ry, rz);
        dst[9 * t + 3 * k + 0] = rx;
        dst[9 * t + 3 * k + 1] = ry;
        dst[9 * t + 3 * k + 2] = rz;
      }
    }
  }
  __syncthreads();  // (B)
  if (threadIdx.x == 0 && !direct) s_misc[4] = cur + total;
}

// Final flush of the LDS triangle stage (all threads of the workgroup).
__device__ __forceinline__ void mc_final_flush(float* s_stage, unsigned* s_misc, unsigned long long* s_base,
                                               float* __restrict__ tris, uint64_t tri_cap, MeshCounters* __restrict__ ctr) {
  __syncthreads();
  const unsigned cur = s_misc[4];
  if (cur) {
    if (threadIdx.x == 0) *s_base = atomicAdd(&ctr->n_tris, (unsigned long long)cur);
    __syncthreads();
    const unsigned long long fb = *s_base;
    if (fb + cur <= tri_cap) {
      float* dst = tris + fb * 9;
      for (unsigned k = threadIdx.x; k < cur * 9; k += BLOCK) dst[k] = s_stage[k];
    } else if (threadIdx.x == 0) {
      ctr->overflow = 1ull;
    }
  }
}


// Balanced marching-cubes emission of one workgroup pass: ONE TRIANGLE PER LANE instead of one cube per lane.
// Every lane brings NC cubes (index[c] = 8-bit inside mask; 0 and 255 give no triangles). The block prefix sum of the
// per-cube triangle counts gives every triangle a slot; the cubes write an owner list (cube id = c*BLOCK + thread,
// triangle number) into LDS, and lane t builds triangle t from its owner's data: corner(id, 0..7) = corner distances,
// origin(id, x0, y0, z0) = min corner (max corner = min + res, as Box{origin, origin+size}). With ~1 cube in 5 cut by
// the surface and up to five triangles per cube, the cube-per-lane loop of mc_emit_block keeps a wave busy for five
// rounds on behalf of a few lanes; here all lanes work for ceil(total/BLOCK) rounds.
// Triangles are staged in LDS (`cur` = staged count: block-uniform, held in a register by every thread) and flushed
// coalesced with ONE append on the global counter per STAGE triangles.
// Block-uniform control flow: every thread of the workgroup calls this together. Ends with a barrier, so the caller
// may overwrite whatever corner()/origin() read. LDS: s_owner[5*BLOCK*NC] u16, s_index[BLOCK*NC] u8.
template <int STAGE>
__device__ __forceinline__ void mc_stage_flush(float* s_stage, unsigned long long* s_base, unsigned& cur, float* __restrict__ tris,
                                               uint64_t tri_cap, MeshCounters* __restrict__ ctr) {
#ifdef GSDF_EXP_NO_FLUSH  // developer experiment: emission without the global append (timing only)
  __syncthreads();
  cur = 0;
  return;
#endif
  if (threadIdx.x == 0) *s_base = atomicAdd(&ctr->n_tris, (unsigned long long)cur);
  __syncthreads();
  const unsigned long long fb = *s_base;
  if (fb + cur <= tri_cap) {
    float* dst = tris + fb * 9;
    for (unsigned k = threadIdx.x; k < cur * 9; k += BLOCK) dst[k] = s_stage[k];
  } else if (threadIdx.x == 0) {
    ctr->overflow = 1ull;  // the counter keeps counting: the host learns the exact size and reruns
  }
  __syncthreads();
  cur = 0;
}

template <int NC, int STAGE, typename Corner, typename Origin>
__device__ __forceinline__ void mc_emit_balanced(const unsigned (&index)[NC], uint16_t* s_owner, uint8_t* s_index, const int8_t* s_tri,
                                                 float* s_stage, unsigned* s_misc, unsigned long long* s_base, unsigned& cur,
                                                 float res, Corner corner, Origin origin, float* __restrict__ tris, uint64_t tri_cap,
                                                 MeshCounters* __restrict__ ctr) {
  constexpr unsigned ID_BITS = NC == 1 ? 8 : (NC == 2 ? 9 : (NC <= 4 ? 10 : 11));
  static_assert(NC <= 8, "owner entries are 16 bits: 3 bits of triangle number + 11 bits of cube id");
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned nt[NC], ntl = 0;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    unsigned n = 0;
    if (index[c]) {  // row 0 is empty anyway; saves the LDS walk for the common case
      const int8_t* row = s_tri + index[c] * 16;
      while (n < 5 && row[3 * n] >= 0) n++;
    }
    nt[c] = n;
    ntl += n;
  }
  if (!__syncthreads_or((int)ntl)) return;  // no triangles anywhere in this pass (block-uniform)
  unsigned incl = ntl;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned u = __shfl_up(incl, off, 64);
    if (lane >= (unsigned)off) incl += u;
  }
  if (lane == 63) s_misc[wave] = incl;
  __syncthreads();
  // block-uniform values read back from LDS: pin them to SGPRs (the compiler cannot know they are uniform)
  const unsigned w0 = __builtin_amdgcn_readfirstlane(s_misc[0]), w1 = __builtin_amdgcn_readfirstlane(s_misc[1]),
                 w2 = __builtin_amdgcn_readfirstlane(s_misc[2]), w3 = __builtin_amdgcn_readfirstlane(s_misc[3]);
  const unsigned total = w0 + w1 + w2 + w3;
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
  unsigned first = (wave_u > 0 ? w0 : 0u) + (wave_u > 1 ? w1 : 0u) + (wave_u > 2 ? w2 : 0u) + (incl - ntl);
#pragma unroll
  for (int c = 0; c < NC; c++) {
    if (nt[c]) {
      const unsigned id = (unsigned)c * BLOCK + threadIdx.x;
      s_index[id] = (uint8_t)index[c];
      for (unsigned k = 0; k < nt[c]; k++) s_owner[first + k] = (uint16_t)(id | (k << ID_BITS));
      first += nt[c];
    }
  }
  __syncthreads();
  for (unsigned done = 0; done < total;) {  // block-uniform
    const unsigned room = STAGE - cur, left = total - done;
    const unsigned n = left < room ? left : room;
#ifndef GSDF_EXP_NO_BUILD  // developer experiment: emission without building the triangles (timing only)
    for (unsigned t = threadIdx.x; t < n; t += BLOCK) {
      const unsigned o = s_owner[done + t];
      const unsigned k = o >> ID_BITS, id = o & ((1u << ID_BITS) - 1u);
      float x0, y0, z0;
      origin(id, x0, y0, z0);
      const float x1 = x0 + res, y1 = y0 + res, z1 = z0 + res;
      const int8_t* row = s_tri + (unsigned)s_index[id] * 16 + 3 * k;
      float* dst = s_stage + (size_t)(cur + t) * 9;
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int e = row[2 - j];  // reversed winding (marchcubes.go:64-68)
        const unsigned ca = GSDF_MC_PAIR_A(e), cb = GSDF_MC_PAIR_B(e);
        const bool ax = ((ca ^ (ca >> 1)) & 1u) != 0u, ay = ((ca >> 1) & 1u) != 0u, az = ((ca >> 2) & 1u) != 0u;
        const bool bx = ((cb ^ (cb >> 1)) & 1u) != 0u, by = ((cb >> 1) & 1u) != 0u, bz = ((cb >> 2) & 1u) != 0u;
        float rx, ry, rz;
        mc_interp(ax ? x1 : x0, ay ? y1 : y0, az ? z1 : z0, bx ? x1 : x0, by ? y1 : y0, bz ? z1 : z0, corner(id, ca), corner(id, cb),
                  rx, ry, rz);
        dst[3 * j + 0] = rx;
        dst[3 * j + 1] = ry;
        dst[3 * j + 2] = rz;
      }
    }
#endif
    cur += n;
    done += n;
    __syncthreads();
    if (cur == STAGE) mc_stage_flush<STAGE>(s_stage, s_base, cur, tris, tri_cap, ctr);
  }
}

// Leaf kernel: one lane per leaf cube of every surviving level-lq cube (64 leaves of a level-3 cube
// = one wave). Corner 0 first; the wave runs the other 7 corners only if some lane passes the
// reference's |d0| <= 2*sqrt3*res test (marchcubes.go:20-23). Marching cubes reads the triangle
// table from LDS; triangles are staged in LDS and flushed with ONE global atomic per flush
// (a single counter word saturates at ~88 atomics/us on MI355X, so per-wave appends do not scale).
// LDS: [max(nslots*K, LEAF_MIN_COLS) floats per lane | tri table 256x16 i8 | TRI_STAGE*9 floats | 8 words]. The lane's 8 corner
// distances reuse the interpreter's slot columns: the distances of the earlier passes ride in registers until the
// last pass has finished with the slots, then all 8 are stored for marching cubes' dynamically indexed reads.
template <int K, int WAVES>
__global__ void __launch_bounds__(BLOCK, WAVES) leaf_kernel(const uint32_t* __restrict__ code_g, const Cube* __restrict__ cubes,
                                                     unsigned long long cube_cap, int lq, int nslots, float ox, float oy, float oz,
                                                     float res, float* __restrict__ tris, uint64_t tri_cap,
                                                     MeshCounters* __restrict__ ctr) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  float* vslot = lds;  // 8 per-lane corner distances, written after the last interpreter pass (aliases the slots)
  // columns 8..10: the lane's cube origin; columns 11..13 (3 KB, block-shared): owner list + cube indices of the
  // balanced emission -- all of it aliases slot columns, which are idle between the last evaluation and the barrier
  // that ends the emission
  uint16_t* s_owner = (uint16_t*)(g_smem + 11 * BLOCK);  // [5 * BLOCK]
  uint8_t* s_index = (uint8_t*)(s_owner + 5 * BLOCK);    // [BLOCK]
  int8_t* s_tri = (int8_t*)(g_smem + (size_t)(nslots * K > LEAF_MIN_COLS ? nslots * K : LEAF_MIN_COLS) * BLOCK);
  float* s_stage = (float*)(s_tri + 256 * 16);
  unsigned* s_misc = (unsigned*)(s_stage + TRI_STAGE * 9);  // [0..3] wave sums
  unsigned long long* s_base = (unsigned long long*)(s_misc + 6);
  for (int k = threadIdx.x; k < 256 * 16; k += BLOCK) s_tri[k] = GSDF_MC_TRI[k >> 4][k & 15];
  __syncthreads();
  unsigned cur = 0;  // triangles in the LDS stage (block-uniform)

  const float cubeDiag = 2 * 1.73205080757f * res;  // marchcubes.go:19
  const int sh = lq - 1;
  unsigned long long n_cubes = uniform_u64(ctr->n_level[lq]);  // survivors of the last prune level (device-side count)
  if (n_cubes > cube_cap) n_cubes = cube_cap;                  // queue overflowed: host reruns with larger queues
  const uint64_t n_leaves = uniform_u64(n_cubes << (3 * sh));  // wave-uniform: keep it in SGPRs (the clamp above is a per-lane select otherwise)
  unsigned my_active = 0, my_cont = 0;  // per wave, < 2^32: a workgroup visits at most 2^32 / BLOCK iterations
  const uint64_t step = (uint64_t)gridDim.x * BLOCK;
  for (uint64_t base = (uint64_t)blockIdx.x * BLOCK; base < n_leaves; base += step) {
    const uint64_t i = base + threadIdx.x;
    const bool valid = i < n_leaves;
    Cube lf = {0, 0, 0, 0};
    if (valid) {
      const Cube pc = cubes[i >> (3 * sh)];
      const unsigned l = (unsigned)(i & ((1u << (3 * sh)) - 1u));
      const unsigned m = (1u << sh) - 1u;
      lf.x = (uint16_t)((pc.x << sh) + (l & m));
      lf.y = (uint16_t)((pc.y << sh) + ((l >> sh) & m));
      lf.z = (uint16_t)((pc.z << sh) + ((l >> (2 * sh)) & m));
    }
    const float x0 = ox + res * (float)lf.x, y0 = oy + res * (float)lf.y, z0 = oz + res * (float)lf.z;
    const float x1 = x0 + res, y1 = y0 + res, z1 = z0 + res;  // Box max = origin + size
    // Single interpreter call site, K corners per pass (corner 0 is in the first pass); the wave goes on
    // to the remaining corners only if some lane passes the reference's corner-0 test.
    unsigned index = 0;
    bool pass = false;
    float dall[8];  // distances in evaluation order; shifted so that the final contents sit at static positions
#pragma unroll
    for (int j = 0; j < 8; j++) dall[j] = 0.f;
#pragma unroll 1
    for (unsigned c0 = 0; c0 < 8; c0 += K) {
      P3 pk[K];
      float dk[K];
      // Corner order {0,4,1,5 | 3,7,2,6}: consecutive points share x,y and (K = 4) points j, j+2 share z, which is
      // what the interpreter's PAIRED mode needs to compute hypot/atan2(x,y) and twist sin/cos(z) once per pair.
#pragma unroll
      for (int kp = 0; kp < K; kp++) {
        const unsigned c = (0x62735140u >> (4u * (c0 + kp))) & 7u;
        pk[kp].x = ((c ^ (c >> 1)) & 1u) ? x1 : x0;
        pk[kp].y = ((c >> 1) & 1u) ? y1 : y0;
        pk[kp].z = ((c >> 2) & 1u) ? z1 : z0;
      }
      gsdf_dev::sdf_eval<K, true>(code, pk, dk, lds, BLOCK, /*brick=*/sh == 2);  // lq == 3: one wave = one 4x4x4 brick
#pragma unroll
      for (int j = 0; j < 8 - K; j++) dall[j] = dall[j + K];  // static shift register: no dynamic register index
#pragma unroll
      for (int kp = 0; kp < K; kp++) {
        const unsigned c = (0x62735140u >> (4u * (c0 + kp))) & 7u;
        dall[8 - K + kp] = dk[kp];
        index |= (dk[kp] < 0.f ? 1u : 0u) << c;
      }
      if (c0 == 0) {
        pass = valid && (dm::absf(dk[0]) <= cubeDiag);
        const unsigned long long pmask = __ballot(pass);
        if (pmask == 0ull) break;  // wave-uniform
        const unsigned long long vmask = __ballot(valid);
        // wave-uniform counters (every lane adds the same scalar): they live in SGPRs, not in four VGPRs
        my_active += (unsigned)__builtin_popcountll(pmask);
        my_cont += (unsigned)__builtin_popcountll(vmask);
      }
    }
    if (!pass || index == 255u) index = 0;
#ifdef GSDF_EXP_NO_EMIT  // developer experiment (GSDF_HIP_SPEC_FLAGS=-DGSDF_EXP_NO_EMIT): evaluation cost alone
    index = 0;
#endif
    // after the last pass dall[j] is the distance of corner order[j] (an early exit leaves index == 0: nothing is read)
    if (index) {
#pragma unroll
      for (int j = 0; j < 8; j++) vslot[((0x62735140u >> (4u * j)) & 7u) * BLOCK] = dall[j];
      lds[8 * BLOCK] = x0;
      lds[9 * BLOCK] = y0;
      lds[10 * BLOCK] = z0;
    }
    const unsigned index1[1] = {index};
    mc_emit_balanced<1, TRI_STAGE>(
        index1, s_owner, s_index, s_tri, s_stage, s_misc, s_base, cur, res,
        [&](unsigned id, unsigned cc) { return g_smem[cc * BLOCK + id]; },
        [&](unsigned id, float& ax, float& ay, float& az) {
          ax = g_smem[8 * BLOCK + id];
          ay = g_smem[9 * BLOCK + id];
          az = g_smem[10 * BLOCK + id];
        },
        tris, tri_cap, ctr);
  }
  __syncthreads();
  if (cur) mc_stage_flush<TRI_STAGE>(s_stage, s_base, cur, tris, tri_cap, ctr);
  // statistics: two atomics per workgroup, not per wave (they share the L2 atomic unit with the triangle appends)
  __syncthreads();
  unsigned* s_stat = (unsigned*)s_stage;
  if ((threadIdx.x & 63u) == 0u) { s_stat[2 * (threadIdx.x >> 6)] = my_active; s_stat[2 * (threadIdx.x >> 6) + 1] = my_cont; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long a = (unsigned long long)s_stat[0] + s_stat[2] + s_stat[4] + s_stat[6];
    const unsigned long long c = (unsigned long long)s_stat[1] + s_stat[3] + s_stat[5] + s_stat[7];
    if (c) { atomicAdd(&ctr->n_active, a); atomicAdd(&ctr->n_cont, c); }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Two-kernel leaf phase (default): leaf_eval_kernel evaluates, march_records_kernel builds the triangles.
//
// The fused leaf_kernel above spends 45 % of its time in the emission (block scan, owner list, LDS stage, four workgroup
// barriers per pass and a global append every ~1.4 passes): the four waves of a workgroup stall together at every barrier
// and at the atomic's round trip, with only 3-4 waves per SIMD to cover for them. Here the evaluating kernel has NO
// barrier and NO atomic in its loop -- its waves are independent -- and leaves, per 64-leaf block (one wave pass), the leaves
// the surface cuts as compact records in HBM:
//     hdr[block]                    = number of records (0..64) | number of triangles << 8
//     rec[block][r][c], c < 10      = the block's records side by side (r = rank among the wave's cut lanes): 8 corner distances
//                                     (corner order 0..7), leaf x | y << 16, leaf z | index << 16 (40 B per cut leaf, ~16 % of
//                                     the leaves). A block holds ~10 records: 420 contiguous bytes -- four cache lines; a
//                                     column-major block (tried first) spread them over ten lines and made the marching
//                                     kernel fetch 2.2x the bytes it used.
// Sized for the worst case (64 records per block: no overflow path); only the cut leaves' lines are ever touched.
// march_records_kernel then runs marching cubes over the records alone (see there): it needs no append counter, because the
// evaluating kernel also leaves the triangle count of every block (hdr, bits 8..) and the sums of both counts per group of
// MARCH_GROUP blocks (psum).
// Same float operations on the same values as the fused kernel: the leaf origin is recomputed from the stored leaf
// coordinates by the expression the evaluation used.
// ---------------------------------------------------------------------------------------------------------------------
#define REC_WORDS 10            // dwords per record
#define REC_BLOCK (64 * REC_WORDS)  // dwords per 64-leaf block
#define MARCH_GROUP 64              // blocks per entry of the group sums (records, triangles) the evaluating kernel accumulates

// UCUBE: lq == 3 (every mesh of three levels or more): the 64 leaves of a wave pass are one level-3 cube.
// NTLDS: the triangles-per-case table sits in LDS behind the interpreter's columns; false when exactly those 256 bytes would
// cost a workgroup per CU (the host decides): the counts are then read from the table in global memory. A template argument
// and not a run-time flag: a select between an LDS and a global load makes the compiler form a flat pointer, and ROCm
// 7.0-7.2's backend then dies on some trees ("Illegal instruction detected ... V_CMP_NE_U32_e32 0, $src_shared_base").
template <int K, int WAVES, bool UCUBE = true, bool NTLDS = true>
__global__ void __launch_bounds__(BLOCK, WAVES) leaf_eval_kernel(const uint32_t* __restrict__ code_g, const Cube* __restrict__ cubes,
                                                          unsigned long long cube_cap, int lq, int nslots, float ox, float oy, float oz,
                                                          float res, uint32_t* __restrict__ hdr, uint32_t* __restrict__ rec,
                                                          unsigned long long* __restrict__ psum, unsigned long long n_blocks_cap,
                                                          MeshCounters* __restrict__ ctr) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  // triangles per marching-cubes case, behind the interpreter's columns (256 B)
  uint8_t* s_nt = (uint8_t*)(g_smem + (size_t)(nslots * K > 8 ? nslots * K : 8) * BLOCK);  // (8 rows at least: the brick's distances)
  const int sh = lq - 1;
  // The three loads a wave starts with -- table byte, cube count, first cube -- are issued together (one trip to memory, not
  // three in a row: a workgroup lives for ~5 passes only). The first cube is read before the count is known: its index is
  // clamped into the queue, and the pass is skipped below if the count says so.
  const uint8_t nt0 = NTLDS ? GSDF_MC_NTRI[threadIdx.x] : (uint8_t)0;
  unsigned long long cw_first = 0ull;
  if (UCUBE) {
    uint64_t ci = uniform_u64(((uint64_t)blockIdx.x * BLOCK + (threadIdx.x & ~63u)) >> (3 * sh));
    if (ci >= cube_cap) ci = cube_cap - 1;
    cw_first = *(const unsigned long long*)(cubes + ci);
  }
  unsigned long long n_cubes = ctr->n_level[lq];  // survivors of the last prune level (device-side count)
  cw_first = uniform_u64(cw_first);
  n_cubes = uniform_u64(n_cubes);
  if (NTLDS) {
    s_nt[threadIdx.x] = nt0;
    __syncthreads();
  }
  const float cubeDiag = 2 * 1.73205080757f * res;  // marchcubes.go:19
  if (n_cubes > cube_cap) n_cubes = cube_cap;                  // queue overflowed: host reruns with larger queues
  const uint64_t n_leaves = uniform_u64(n_cubes << (3 * sh));
  unsigned my_active = 0, my_cont = 0, my_cut = 0;  // wave-uniform (SGPRs)
  const uint64_t step = (uint64_t)gridDim.x * BLOCK;
  // UCUBE: the 64 leaves of a wave pass belong to ONE cube (and n_leaves is a multiple of 64), so the cube is a scalar load --
  // issued one pass ahead: a wave has ~5 passes and the load is a trip to L2/HBM it would otherwise sit out at every start
  auto cube_word = [&](uint64_t b) -> unsigned long long {
    const uint64_t li = uniform_u64(b + (threadIdx.x & ~63u));
    if (li >= n_leaves) return 0ull;
    return *(const unsigned long long*)(cubes + (li >> (3 * sh)));
  };
  unsigned long long cw_next = cw_first;
  for (uint64_t base = (uint64_t)blockIdx.x * BLOCK; base < n_leaves; base += step) {
    const uint64_t i = base + threadIdx.x;
    const bool valid = UCUBE ? uniform_u64(base + (threadIdx.x & ~63u)) < n_leaves : i < n_leaves;
    Cube lf = {0, 0, 0, 0};
    const unsigned long long cw = cw_next;
    if (UCUBE) {
      cw_next = cube_word(base + step);
      if (!valid) continue;  // wave-uniform: nothing of this pass is read by anyone
    }
    if (valid) {
      Cube pc;
      if (UCUBE) { pc.x = (uint16_t)cw; pc.y = (uint16_t)(cw >> 16); pc.z = (uint16_t)(cw >> 32); pc.w = 0; }
      else pc = cubes[i >> (3 * sh)];
      const unsigned l = (unsigned)(i & ((1u << (3 * sh)) - 1u));
      const unsigned m = (1u << sh) - 1u;
      lf.x = (uint16_t)((pc.x << sh) + (l & m));
      lf.y = (uint16_t)((pc.y << sh) + ((l >> sh) & m));
      lf.z = (uint16_t)((pc.z << sh) + ((l >> (2 * sh)) & m));
    }
    unsigned index = 0;
    bool pass = false;
    float dc[8];  // the leaf's corner distances, by corner number
    if (UCUBE) {
      // COLUMN BRICK. The 512 evaluations of a brick -- 64 leaves x 8 corners -- are the product of eight x, eight y and eight z
      // coordinates (leaf i of an axis contributes min_i = origin + res*i and max_i = min_i + res, Box{origin, origin+size}):
      // the lane is the (x, y) COLUMN (lane & 7, lane >> 3), its points the eight z. All points of a lane enter the evaluator
      // with the same x, y registers (SHARE = 2), so every subexpression of x and y alone is computed once per lane instead of
      // once per point -- flagged hypot / atan2 in the interpreter, and in the specialised build whatever the compiler's value
      // numbering finds (2-D profiles under an extrusion, sector folds of circular arrays, ...). Same points, same operations on
      // the same values, same bits as one leaf per lane; only the assignment of points to lanes differs. The distances then
      // change hands through the wave's own (now idle) interpreter columns: leaf (i, j, k) = lane i + 4j + 16k reads corner
      // (cx, cy, cz) from column (2i + cx, 2j + cy), row 2k + cz.
      const unsigned lane = threadIdx.x & 63u;
      const unsigned bx = ((unsigned)(cw & 0xffffu)) << 2, by = ((unsigned)((cw >> 16) & 0xffffu)) << 2, bz = ((unsigned)((cw >> 32) & 0xffffu)) << 2;
      const float xa = ox + res * (float)(uint16_t)(bx + ((lane & 7u) >> 1)), ya = oy + res * (float)(uint16_t)(by + (lane >> 4));
      const float px = (lane & 1u) ? xa + res : xa, py = (lane & 8u) ? ya + res : ya;
      float dall[8];  // distances of rows 0..7; static shift register
#pragma unroll
      for (int j = 0; j < 8; j++) dall[j] = 0.f;
#ifdef GSDF_EXP_UNROLL_PASSES  // developer experiment: both passes of a column brick in one body, so that what depends on x and y alone is computed once
#pragma unroll
#else
#pragma unroll 1
#endif
      for (unsigned c0 = 0; c0 < 8; c0 += K) {
        P3 pk[K];
        float dk[K];
#pragma unroll
        for (int kp = 0; kp < K; kp++) {
          const unsigned r = c0 + kp;  // wave-uniform
          const float za = oz + res * (float)(uint16_t)(bz + (r >> 1));
          pk[kp].x = px;
          pk[kp].y = py;
          pk[kp].z = (r & 1u) ? za + res : za;
        }
        gsdf_dev::sdf_eval<K, 2>(code, pk, dk, lds, BLOCK, /*brick=*/true);
#pragma unroll
        for (int j = 0; j < 8 - K; j++) dall[j] = dall[j + K];
#pragma unroll
        for (int kp = 0; kp < K; kp++) dall[8 - K + kp] = dk[kp];
      }
      float* D = g_smem + (threadIdx.x & ~63u);  // rows of BLOCK floats; this wave's 64 columns of each
#pragma unroll
      for (int r = 0; r < 8; r++) D[r * BLOCK + lane] = dall[r];
      __builtin_amdgcn_wave_barrier();  // (the wave's LDS operations execute in order; this only pins the compiler's schedule)
      const unsigned li = lane & 3u, lj = (lane >> 2) & 3u, lk = lane >> 4;
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const unsigned cx = (c ^ (c >> 1)) & 1u, cy = (c >> 1) & 1u, cz = (c >> 2) & 1u;
        dc[c] = D[(2u * lk + cz) * BLOCK + (2u * lj + cy) * 8u + 2u * li + cx];
        index |= (dc[c] < 0.f ? 1u : 0u) << c;
      }
      pass = dm::absf(dc[0]) <= cubeDiag;
      my_active += (unsigned)__builtin_popcountll(__ballot(pass));
      my_cont += 64u;
    } else {
      const float x0 = ox + res * (float)lf.x, y0 = oy + res * (float)lf.y, z0 = oz + res * (float)lf.z;
      const float x1 = x0 + res, y1 = y0 + res, z1 = z0 + res;  // Box max = origin + size
      float dall[8];  // distances in evaluation order {0,4,1,5,3,7,2,6}; static shift register
#pragma unroll
      for (int j = 0; j < 8; j++) dall[j] = 0.f;
#pragma unroll 1
      for (unsigned c0 = 0; c0 < 8; c0 += K) {
        P3 pk[K];
        float dk[K];
#pragma unroll
        for (int kp = 0; kp < K; kp++) {
          const unsigned c = (0x62735140u >> (4u * (c0 + kp))) & 7u;
          pk[kp].x = ((c ^ (c >> 1)) & 1u) ? x1 : x0;
          pk[kp].y = ((c >> 1) & 1u) ? y1 : y0;
          pk[kp].z = ((c >> 2) & 1u) ? z1 : z0;
        }
        gsdf_dev::sdf_eval<K, true>(code, pk, dk, lds, BLOCK, /*brick=*/false);
#pragma unroll
        for (int j = 0; j < 8 - K; j++) dall[j] = dall[j + K];
#pragma unroll
        for (int kp = 0; kp < K; kp++) {
          const unsigned c = (0x62735140u >> (4u * (c0 + kp))) & 7u;
          dall[8 - K + kp] = dk[kp];
          index |= (dk[kp] < 0.f ? 1u : 0u) << c;
        }
        if (c0 == 0) {
          pass = valid && (dm::absf(dk[0]) <= cubeDiag);
          const unsigned long long pmask = __ballot(pass);
          if (pmask == 0ull) break;  // wave-uniform
          const unsigned long long vmask = __ballot(valid);
          my_active += (unsigned)__builtin_popcountll(pmask);
          my_cont += (unsigned)__builtin_popcountll(vmask);
        }
      }
      // dall[j] is the distance of corner order[j], order = {0,4,1,5,3,7,2,6}
      dc[0] = dall[0]; dc[1] = dall[2]; dc[2] = dall[6]; dc[3] = dall[4]; dc[4] = dall[1]; dc[5] = dall[3]; dc[6] = dall[7]; dc[7] = dall[5];
    }
    const bool cut = pass && index != 0u && index != 255u;
    // compact the cut leaves of this wave's block: rank among the cut lanes, one header word per block
    const unsigned long long cm = __ballot(cut);
    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
    const uint64_t blk = uniform_u64((base + (uint64_t)(threadIdx.x & ~63u)) >> 6);  // block = wave pass = 64 consecutive leaves
    my_cut += (unsigned)__builtin_popcountll(cm);
    if (blk < n_blocks_cap) {
      // header word: records | triangles << 8; the same pair is added to the sum of the block's group of MARCH_GROUP
      // blocks (low / high half of one 64-bit word: a fire-and-forget atomic, one per wave pass the surface cuts), from which
      // march_records_kernel derives every workgroup's share of the records and its triangles' place in the output
      unsigned ntri = 0;
#ifndef GSDF_EXP_NO_NTRI  // developer experiments (GSDF_HIP_SPEC_FLAGS=-DGSDF_EXP_NO_NTRI / _NO_PSUM): what the counts cost (timing only)
      if (cm != 0ull) {  // wave-uniform
        unsigned nt = 0u;  // 0..5
        if (NTLDS) { if (cut) nt = (unsigned)s_nt[index]; }
        else { if (cut) nt = (unsigned)GSDF_MC_NTRI[index]; }
        ntri = (unsigned)__builtin_popcountll(__ballot((nt & 1u) != 0u)) + 2u * (unsigned)__builtin_popcountll(__ballot((nt & 2u) != 0u)) +
               4u * (unsigned)__builtin_popcountll(__ballot((nt & 4u) != 0u));
      }
#endif
      if ((threadIdx.x & 63u) == 0u) {
        const uint32_t nrec = (uint32_t)__builtin_popcountll(cm);
        hdr[blk] = nrec | (ntri << 8);
#ifndef GSDF_EXP_NO_PSUM
        if (nrec) atomicAdd(&psum[blk / MARCH_GROUP], (unsigned long long)nrec | ((unsigned long long)ntri << 32));
#endif
      }
      if (cut) {
        uint2* w = (uint2*)(rec + blk * REC_BLOCK + rank * REC_WORDS);  // 40-byte records: 8-byte aligned, five 8-byte stores
        w[0] = make_uint2(__float_as_uint(dc[0]), __float_as_uint(dc[1]));
        w[1] = make_uint2(__float_as_uint(dc[2]), __float_as_uint(dc[3]));
        w[2] = make_uint2(__float_as_uint(dc[4]), __float_as_uint(dc[5]));
        w[3] = make_uint2(__float_as_uint(dc[6]), __float_as_uint(dc[7]));
        w[4] = make_uint2((uint32_t)lf.x | ((uint32_t)lf.y << 16), (uint32_t)lf.z | (index << 16));
      }
    }
  }
  // statistics: three atomics per workgroup
  unsigned* s_stat = (unsigned*)g_smem;
  __syncthreads();  // everyone is done with the interpreter columns
  if ((threadIdx.x & 63u) == 0u) {
    s_stat[3 * (threadIdx.x >> 6)] = my_active; s_stat[3 * (threadIdx.x >> 6) + 1] = my_cont; s_stat[3 * (threadIdx.x >> 6) + 2] = my_cut;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long a = (unsigned long long)s_stat[0] + s_stat[3] + s_stat[6] + s_stat[9];
    const unsigned long long c = (unsigned long long)s_stat[1] + s_stat[4] + s_stat[7] + s_stat[10];
    const unsigned long long u = (unsigned long long)s_stat[2] + s_stat[5] + s_stat[8] + s_stat[11];
    if (c) { atomicAdd(&ctr->n_active, a); atomicAdd(&ctr->n_cont, c); }
    if (u) atomicAdd(&ctr->n_cut, u);
  }
}

// Inclusive prefix sum over the workgroup (thread order) of a 64-bit value; *total = the workgroup's sum. s_w: 4 words of LDS.
__device__ __forceinline__ unsigned long long block_scan_u64(unsigned long long v, unsigned long long* s_w, unsigned long long* total) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned lo = __shfl_up((unsigned)incl, off, 64), hi = __shfl_up((unsigned)(incl >> 32), off, 64);
    if (lane >= (unsigned)off) incl += ((unsigned long long)hi << 32) | lo;
  }
  __syncthreads();  // s_w may still be read from an earlier call
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  const unsigned long long a = s_w[0], b = s_w[1], c = s_w[2], d = s_w[3];
  *total = uniform_u64(a + b + c + d);
  return incl + (wave > 0 ? a : 0ull) + (wave > 1 ? b : 0ull) + (wave > 2 ? c : 0ull);
}

// Marching cubes over the cut-leaf records. NO atomic, NO staging.
//  * Where things go: the evaluating kernel left, per group of MARCH_GROUP blocks, the number of records and of triangles
//    (psum); every workgroup sums those (a few KB from L2), takes an equal share of the RECORDS -- a contiguous range of
//    blocks, cut at block granularity -- and knows from the same sums where its first triangle goes. Triangles appear in
//    block order, record order, table order (a pure function of the survivor queue's order).
//  * Who computes what: records are taken 256 at a time (one per lane, 8 distances + origin into LDS columns); a prefix sum of
//    their triangle counts gives an owner list (triangle -> record, table row); then ONE OUTPUT VERTEX PER LANE: lane k of a
//    round computes vertex k % 3 of triangle k / 3 and stores its 12 bytes at out*36 + 12 k -- one store instruction of a wave
//    is 768 contiguous bytes, nothing is staged, and the rounds of a chunk are independent of each other (no barrier between
//    them). (One output FLOAT per lane -- 256-byte stores, the edge parameter computed three times -- was slower: 0.127 ms.)
//  (History: the first version appended LDS stages of 896 triangles through the one counter word, ~88 appends/us: three
//  workgroups per CU, and a static deal of 256-block passes of which a workgroup got one or two -- it ran for two pass times
//  with half its slots idle in the second: 0.158 ms. Known offsets + equal shares + a 512-triangle stage: 0.115 ms.)
// LDS: [11 record columns of BLOCK floats | owner list 5*BLOCK u32 | tri table | prefix BLOCK+1 | misc] = 21.7 KB: 7 workgroups per CU
__global__ void __launch_bounds__(BLOCK, 7) march_records_kernel(const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ rec,
                                                              const unsigned long long* __restrict__ psum,
                                                              unsigned long long n_blocks_cap, int lq, float ox, float oy, float oz,
                                                              float res, float* __restrict__ tris, uint64_t tri_cap,
                                                              MeshCounters* __restrict__ ctr) {
  float* s_col = g_smem;                                       // [BLOCK][11]: 8 distances + origin of the chunk's records (odd stride:
                                                               // the values of one record, read together by neighbouring lanes, sit in 11 banks)
  uint32_t* s_own = (uint32_t*)(s_col + 11 * BLOCK);           // [5 * BLOCK] triangle -> table offset (index*16 + 3*number) | record << 12
  int8_t* s_tri = (int8_t*)(s_own + 5 * BLOCK);
  unsigned* s_pre = (unsigned*)(s_tri + 256 * 16);             // [BLOCK + 1] exclusive prefix of the pass's record counts
  unsigned* s_misc = s_pre + BLOCK + 1;                        // [0..3] wave sums of the triangle counts, [4..7] of the record counts
  unsigned long long* s_u64 = (unsigned long long*)(((uintptr_t)(s_misc + 8) + 7) & ~(uintptr_t)7);  // [0..3] scan, [4..9] found, [10..13] result
  for (int k = threadIdx.x; k < 256 * 4; k += BLOCK) {  // the table by dwords; a row's spare byte 15 takes its triangle count
    uint32_t w = ((const uint32_t*)&GSDF_MC_TRI[0][0])[k];
    if ((k & 3) == 3) w = (w & 0x00ffffffu) | ((uint32_t)GSDF_MC_NTRI[k >> 2] << 24);
    ((uint32_t*)s_tri)[k] = w;
  }
  unsigned long long n_cubes = uniform_u64(ctr->n_level[lq]);
  const uint64_t n_leaves = n_cubes << (3 * (lq - 1));
  uint64_t n_blocks = (n_leaves + 63) >> 6;
  if (n_blocks > n_blocks_cap) n_blocks = n_blocks_cap;  // (the cube queue overflowed: the host reruns)
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;

  // ---- this workgroup's share of the records, and where its triangles go
  const uint64_t n_grp = (n_blocks + MARCH_GROUP - 1) / MARCH_GROUP;
  const uint64_t per = (n_grp + BLOCK - 1) / BLOCK;  // groups per thread (contiguous)
  uint64_t e0 = (uint64_t)threadIdx.x * per, e1 = e0 + per;
  if (e0 > n_grp) e0 = n_grp;
  if (e1 > n_grp) e1 = n_grp;
  unsigned long long lr = 0, lt = 0;
#pragma unroll 4
  for (uint64_t e = e0; e < e1; e++) {
    const unsigned long long v = psum[e];
    lr += (unsigned)v;
    lt += v >> 32;
  }
  unsigned long long R, T;
  const unsigned long long br = block_scan_u64(lr, s_u64, &R) - lr, bt = block_scan_u64(lt, s_u64, &T) - lt;
  if (R == 0ull) return;  // no surface here (n_tris stays 0)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctr->n_tris = T;
    if (T > tri_cap) ctr->overflow = 1ull;  // the host learns the exact size and reruns
  }
  if (T > tri_cap) return;
  const unsigned long long X0 = R * blockIdx.x / gridDim.x, X1 = R * (blockIdx.x + 1ull) / gridDim.x;  // records [X0, X1)
  if (X0 == X1) return;
  // the group in which the running record count reaches X (X > 0): found by the one thread whose groups straddle it
#pragma unroll
  for (int w = 0; w < 2; w++) {
    const unsigned long long X = w ? X1 : X0;
    if (br < X && X <= br + lr) {
      unsigned long long acc = br, tacc = bt;
      for (uint64_t e = e0; e < e1; e++) {
        const unsigned long long v = psum[e];
        if (acc + (unsigned)v >= X) {
          s_u64[4 + 3 * w] = e; s_u64[5 + 3 * w] = acc; s_u64[6 + 3 * w] = tacc;
          break;
        }
        acc += (unsigned)v;
        tacc += v >> 32;
      }
    }
  }
  __syncthreads();
  // first block whose exclusive record prefix is >= X, and the triangles before it (waves 0 and 1: X0 and X1)
  if (wave < 2) {
    const unsigned long long X = wave ? X1 : X0;
    unsigned long long B = 0, tb = 0;
    if (X != 0ull) {
      const unsigned long long e = s_u64[4 + 3 * wave], acc = s_u64[5 + 3 * wave], tacc = s_u64[6 + 3 * wave];
      const uint64_t b = e * MARCH_GROUP + lane;
      const uint32_t h = b < n_blocks ? hdr[b] : 0u;
      const unsigned nr = h & 255u, ntr = h >> 8;
      unsigned ir = nr, it = ntr;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned ur = __shfl_up(ir, off, 64), ut = __shfl_up(it, off, 64);
        if (lane >= (unsigned)off) { ir += ur; it += ut; }
      }
      const unsigned long long m = __ballot(acc + (ir - nr) >= X);
      if (m != 0ull) {
        const int j = __builtin_ctzll(m);
        B = e * MARCH_GROUP + (unsigned)j;
        tb = tacc + (unsigned)__shfl(it - ntr, j, 64);
      } else {
        B = (e + 1) * MARCH_GROUP;
        tb = tacc + (unsigned)__shfl(it, 63, 64);
      }
    }
    if (lane == 0) { s_u64[10 + 2 * wave] = B; s_u64[11 + 2 * wave] = tb; }
  }
  __syncthreads();
  const uint64_t b_begin = uniform_u64(s_u64[10]);
  uint64_t b_end = uniform_u64(s_u64[12]);
  if (b_end > n_blocks) b_end = n_blocks;
  unsigned long long out = uniform_u64(s_u64[11]);

  for (uint64_t b0 = b_begin; b0 < b_end; b0 += BLOCK) {  // block-uniform
    // exclusive prefix of the record counts of blocks b0 .. b0+255
    const uint64_t b = b0 + threadIdx.x;
    const unsigned nr = b < b_end ? (hdr[b] & 255u) : 0u;
    unsigned incl = nr;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned u = __shfl_up(incl, off, 64);
      if (lane >= (unsigned)off) incl += u;
    }
    if (lane == 63) s_misc[4 + wave] = incl;
    __syncthreads();
    const unsigned p0 = s_misc[4], p1 = s_misc[5], p2 = s_misc[6], p3 = s_misc[7];
    const unsigned wpre = (wave > 0 ? p0 : 0u) + (wave > 1 ? p1 : 0u) + (wave > 2 ? p2 : 0u);
    s_pre[threadIdx.x] = wpre + incl - nr;
    const unsigned Rp = __builtin_amdgcn_readfirstlane(p0 + p1 + p2 + p3);  // records of this pass (block-uniform)
    if (threadIdx.x == 0) s_pre[BLOCK] = Rp;
    __syncthreads();
    // One record per lane in chunks of 256. The next chunk's record is fetched into registers BEFORE the current chunk is
    // marched: a dependent global load is ~2 us.
    uint32_t rw[REC_WORDS];
    auto fetch = [&](unsigned q) {
#pragma unroll
      for (int c = 0; c < REC_WORDS; c++) rw[c] = 0u;
      if (q < Rp) {
        // the block holding record q: largest j with s_pre[j] <= q (blocks without records share their successor's prefix)
        unsigned lo = 0, hi = BLOCK;
#pragma unroll
        for (int it = 0; it < 8; it++) {
          const unsigned mid = (lo + hi) >> 1;
          if (s_pre[mid] <= q) lo = mid; else hi = mid;
        }
        struct __attribute__((packed, aligned(8))) Rec { uint32_t w[REC_WORDS]; };  // 40 bytes, 8-byte aligned: wide loads
#ifdef GSDF_EXP_MARCH_ONE_LINE  // developer experiment: every lane reads the FIRST record of its block (a third of the record traffic)
        const Rec v = *(const Rec*)(rec + (b0 + lo) * REC_BLOCK);
#else
        const Rec v = *(const Rec*)(rec + (b0 + lo) * REC_BLOCK + (q - s_pre[lo]) * REC_WORDS);
#endif
#pragma unroll
        for (int c = 0; c < REC_WORDS; c++) rw[c] = v.w[c];
      }
    };
    fetch(threadIdx.x);
    for (unsigned q0 = 0; q0 < Rp; q0 += BLOCK) {  // block-uniform
      const unsigned q = q0 + threadIdx.x;
      unsigned index = 0;
      if (q < Rp) {
#pragma unroll
        for (int c = 0; c < 8; c++) s_col[threadIdx.x * 11u + c] = __uint_as_float(rw[c]);
        const uint32_t xy = rw[8], zi = rw[9];
        index = zi >> 16;
        // the leaf origin exactly as the evaluating kernel formed it
        s_col[threadIdx.x * 11u + 8u] = ox + res * (float)(xy & 0xffffu);
        s_col[threadIdx.x * 11u + 9u] = oy + res * (float)(xy >> 16);
        s_col[threadIdx.x * 11u + 10u] = oz + res * (float)(zi & 0xffffu);
      }
      fetch(q + BLOCK);  // in flight while this chunk is marched
      // owner list: prefix sum of the records' triangle counts (the table's spare byte holds the row's count)
      const unsigned nt = index ? (unsigned)s_tri[index * 16 + 15] : 0u;
      unsigned ti = nt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned u = __shfl_up(ti, off, 64);
        if (lane >= (unsigned)off) ti += u;
      }
      if (lane == 63) s_misc[wave] = ti;
      __syncthreads();
      const unsigned w0 = __builtin_amdgcn_readfirstlane(s_misc[0]), w1 = __builtin_amdgcn_readfirstlane(s_misc[1]),
                     w2 = __builtin_amdgcn_readfirstlane(s_misc[2]), w3 = __builtin_amdgcn_readfirstlane(s_misc[3]);
      const unsigned total = w0 + w1 + w2 + w3;
      const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
      const unsigned first = (wave_u > 0 ? w0 : 0u) + (wave_u > 1 ? w1 : 0u) + (wave_u > 2 ? w2 : 0u) + (ti - nt);
      for (unsigned k = 0; k < nt; k++) s_own[first + k] = (index * 16u + 3u * k) | (threadIdx.x << 12);
      __syncthreads();
      // one output VERTEX per lane: a wave's store is 768 contiguous bytes
      struct __attribute__((packed, aligned(4))) V3 { float x, y, z; };
      V3* dst = (V3*)(tris + out * 9);
      const unsigned n3 = total * 3u;
#pragma unroll 2
      for (unsigned k = threadIdx.x; k < n3; k += BLOCK) {
        const unsigned t = k / 3u, j = k - 3u * t;
        const uint32_t o = s_own[t];
        const float* col = s_col + (o >> 12) * 11u;
        const int e = s_tri[(o & 4095u) + (2u - j)];  // reversed winding (marchcubes.go:64-68)
        const unsigned ca = GSDF_MC_PAIR_A(e), cb = GSDF_MC_PAIR_B(e);
        const float x0 = col[8], y0 = col[9], z0 = col[10];
        const float x1 = x0 + res, y1 = y0 + res, z1 = z0 + res;  // Box max = origin + size
        const bool ax = ((ca ^ (ca >> 1)) & 1u) != 0u, ay = ((ca >> 1) & 1u) != 0u, az = ((ca >> 2) & 1u) != 0u;
        const bool bx = ((cb ^ (cb >> 1)) & 1u) != 0u, by = ((cb >> 1) & 1u) != 0u, bz = ((cb >> 2) & 1u) != 0u;
        V3 r;
        mc_interp(ax ? x1 : x0, ay ? y1 : y0, az ? z1 : z0, bx ? x1 : x0, by ? y1 : y0, bz ? z1 : z0, col[ca], col[cb], r.x, r.y, r.z);
#ifdef GSDF_EXP_MARCH_NO_STORE  // developer experiment (library built with -D...): the kernel without its output stream (timing only)
        if (r.x == 1.2345678e-30f) dst[k] = r;
#else
        dst[k] = r;
#endif
      }
      out += total;
      __syncthreads();  // the next chunk rewrites the columns, the owner list and s_misc
    }
  }
}

// Leaf kernel with exact corner sharing (level-3 bricks: one wave = one brick of 4x4x4 leaves).
// The reference evaluates 8 corners per leaf: 512 evaluations per brick. Neighbouring leaves share lattice
// planes, but the two coordinate expressions of a plane -- A(i) = O + res*i (min corner of leaf i) and
// B(i) = A(i-1) + res (max corner of leaf i-1) -- are only sometimes the same float (64-73 % of planes at
// resdiv 1600). Per axis the brick therefore has 5..8 bitwise-distinct coordinates (A0, {B1,A1}, {B2,A2},
// {B3,A3}, B4 with equal pairs merged); every distinct point is evaluated ONCE (typically ~6x6x6 = 216
// instead of 512: one 4-points-per-lane pass instead of two) and each leaf corner reads the value of
// exactly the coordinates the reference would have evaluated, so distances, signs and triangles stay
// bit-identical.
// LDS: [nslots*K floats per lane | tri table | triangle stage | misc | 4 x 512 distances | 4 x 24 coordinates].
template <int K, int WAVES>
__global__ void __launch_bounds__(BLOCK, WAVES) leaf_brick_kernel(const uint32_t* __restrict__ code_g, const Cube* __restrict__ cubes,
                                                                  unsigned long long cube_cap, int nslots, float ox, float oy, float oz,
                                                                  float res, float* __restrict__ tris, uint64_t tri_cap,
                                                                  MeshCounters* __restrict__ ctr) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  int8_t* s_tri = (int8_t*)(g_smem + (size_t)(nslots * K) * BLOCK);
  float* s_stage = (float*)(s_tri + 256 * 16);
  unsigned* s_misc = (unsigned*)(s_stage + TRI_STAGE * 9);
  unsigned long long* s_base = (unsigned long long*)(s_misc + 6);
  float* s_D = (float*)(s_base + 1);
  float* s_val = s_D + 4 * 512;
  for (int k = threadIdx.x; k < 256 * 16; k += BLOCK) s_tri[k] = GSDF_MC_TRI[k >> 4][k & 15];
  if (threadIdx.x == 0) s_misc[4] = 0;
  __syncthreads();

  const float cubeDiag = 2 * 1.73205080757f * res;  // marchcubes.go:19
  unsigned long long n_cubes = uniform_u64(ctr->n_level[3]);
  if (n_cubes > cube_cap) n_cubes = cube_cap;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* D = s_D + wave * 512;
  float* val = s_val + wave * 24;
  const float org[3] = {ox, oy, oz};
  unsigned long long my_active = 0, my_points = 0;
  const uint64_t step = (uint64_t)gridDim.x * 4;
  for (uint64_t base = (uint64_t)blockIdx.x * 4; base < n_cubes; base += step) {  // block-uniform trip count
    const uint64_t brick = base + wave;
    const bool bvalid = brick < n_cubes;
    Cube pc = {0, 0, 0, 0};
    if (bvalid) pc = cubes[brick];
    const unsigned pidx[3] = {pc.x, pc.y, pc.z};
    // per-axis mismatch bits m_p (p = 1..3): plane p has two distinct floats
    unsigned mb[3], nax[3];
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
      const unsigned i0 = pidx[ax] * 4u;
      float A[4];
#pragma unroll
      for (int k = 0; k < 4; k++) A[k] = org[ax] + res * (float)(i0 + k);  // CubeOrigin of leaf i0+k
      unsigned m = 0;
#pragma unroll
      for (int k = 1; k < 4; k++) m |= ((A[k - 1] + res) != A[k] ? 1u : 0u) << (k - 1);
      mb[ax] = m;
      nax[ax] = 5u + __builtin_popcount(m);
    }
    if (lane < 12) {  // coordinate table: lane (axis, a) writes A_a and B_{a+1} at their distinct-value slots
      const unsigned ax = lane >> 2, a = lane & 3u;
      const float Aa = org[ax] + res * (float)(pidx[ax] * 4u + a);
      const unsigned u = a + __builtin_popcount(mb[ax] & ((1u << a) - 1u));
      val[ax * 8 + u] = Aa;
      val[ax * 8 + u + 1] = Aa + res;  // Box max = origin + size
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned nx = nax[0], nxy = nax[0] * nax[1], N = nxy * nax[2];
    const float inx = 1.0f / (float)nx, inxy = 1.0f / (float)nxy;
    if (bvalid && lane == 0) my_points += N;
#pragma unroll 1
    for (unsigned t0 = 0; t0 < N; t0 += 64 * K) {  // wave-uniform: 1 pass when N <= 256
      P3 pk[K];
      float dk[K];
#pragma unroll
      for (int kp = 0; kp < K; kp++) {
        unsigned t = t0 + kp * 64 + lane;
        if (t >= N) t = N - 1;  // idle slots re-evaluate the last point (result discarded)
        const unsigned uz = (unsigned)(((float)t + 0.5f) * inxy);
        const unsigned r = t - uz * nxy;
        const unsigned uy = (unsigned)(((float)r + 0.5f) * inx);
        const unsigned ux = r - uy * nx;
        pk[kp] = P3{val[ux], val[8 + uy], val[16 + uz]};
      }
      gsdf_dev::sdf_eval<K>(code, pk, dk, lds, BLOCK);
#pragma unroll
      for (int kp = 0; kp < K; kp++) {
        const unsigned t = t0 + kp * 64 + lane;
        if (t < N) D[t] = dk[kp];
      }
    }
    __builtin_amdgcn_wave_barrier();
    // this lane's leaf (a,b,c) and its corner 0 index in the distinct-point lattice
    const unsigned la = lane & 3u, lb = (lane >> 2) & 3u, lc = lane >> 4;
    const unsigned ux0 = la + __builtin_popcount(mb[0] & ((1u << la) - 1u));
    const unsigned uy0 = lb + __builtin_popcount(mb[1] & ((1u << lb) - 1u));
    const unsigned uz0 = lc + __builtin_popcount(mb[2] & ((1u << lc) - 1u));
    const unsigned tb = ux0 + nx * uy0 + nxy * uz0;
    auto vdist = [&](unsigned cc) { return D[tb + ((cc ^ (cc >> 1)) & 1u) + nx * ((cc >> 1) & 1u) + nxy * ((cc >> 2) & 1u)]; };
    const float x0 = val[ux0], x1 = val[ux0 + 1], y0 = val[8 + uy0], y1 = val[8 + uy0 + 1], z0 = val[16 + uz0], z1 = val[16 + uz0 + 1];
    unsigned index = 0;
#pragma unroll
    for (unsigned cc = 0; cc < 8; cc++) index |= (vdist(cc) < 0.f ? 1u : 0u) << cc;
    const bool pass = bvalid && (dm::absf(vdist(0)) <= cubeDiag);
    const unsigned long long pmask = __ballot(pass);
    if (lane == 0) my_active += (unsigned long long)__builtin_popcountll(pmask);
    if (!pass) index = 0;
    mc_emit_block(index, x0, y0, z0, x1, y1, z1, vdist, s_tri, s_stage, s_misc, s_base, tris, tri_cap, ctr);
  }
  mc_final_flush(s_stage, s_misc, s_base, tris, tri_cap, ctr);
  if (lane == 0 && my_points) {
    atomicAdd(&ctr->n_active, my_active);
    atomicAdd(&ctr->n_points, my_points);
  }
}

// =================================================================================================
// FlatRenderer on device (glrender/flatrenderer.go): the SDF on every corner of the (nx+1)(ny+1)(nz+1) lattice into a
// dense grid in HBM (1.7 GB at npt-flange resdiv 1600: with 288 GB the reference's layout is affordable as it is), then
// marching cubes of every cube out of the grid.
// =================================================================================================
// evalKRange (:146-182): grid[i + sx*(j + sy*k)] = SDF(origin + (i,j,k)*res) for the planes [kfirst, kfirst+nk) of the
// lattice, into a slab whose first plane is kfirst. A lane carries ONE lattice column (i,j) on K consecutive planes, so
// its K points enter the evaluator with bitwise equal x,y (COLUMN mode: every hypot/atan2 of x,y is computed once per
// lane, not once per point); a workgroup pass covers BLOCK columns of one group of K planes, and stores stay coalesced
// (consecutive lanes = consecutive columns of a plane). (i,j) comes from one division per lane and pass.
// Rows of the distance grid start on a 256-byte boundary (pitch = sx rounded up to 64 floats): flat_march_kernel's row loads
// (64 lanes x 4 B) then cover two 128-byte lines instead of straddling three.
#define FLAT_PITCH(sx) (((sx) + 63u) & ~63u)
template <int K, int W = 3>
__global__ void __launch_bounds__(BLOCK, W) flat_grid_kernel(const uint32_t* __restrict__ code_g, float ox, float oy, float oz, float res,
                                                             unsigned sx, unsigned sy, unsigned kfirst, unsigned nk,
                                                             float* __restrict__ grid, unsigned long long* __restrict__ negbits,
                                                             unsigned long long* __restrict__ nearbits) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  const unsigned sxy = sx * sy;                      // < 2^32 (host checks)
  const unsigned tpg = (sxy + BLOCK - 1) / BLOCK;  // passes per plane group
  const float cubeDiag = 2 * 1.73205080757f * res;  // marchcubes.go:19 / flatrenderer.go:207
  const unsigned ngroups = (nk + K - 1) / K;
  const uint64_t npass = (uint64_t)tpg * ngroups;
  for (uint64_t w = blockIdx.x; w < npass; w += gridDim.x) {  // uniform trip count
    const unsigned g = (unsigned)(w / tpg), t = (unsigned)(w - (uint64_t)g * tpg);
    const unsigned col = t * BLOCK + threadIdx.x;
    const unsigned c = col < sxy ? col : sxy - 1;  // padding lanes evaluate a valid point, nothing is stored
    const unsigned j = c / sx, i = c - j * sx;
    const float x = ox + (float)i * res, y = oy + (float)j * res;
    P3 p[K];
    float d[K];
#pragma unroll
    for (int z = 0; z < K; z++) p[z] = P3{x, y, oz + (float)(kfirst + g * K + (unsigned)z) * res};
    gsdf_dev::sdf_eval<K, 2>(code, p, d, lds, BLOCK);
#pragma unroll
    for (int z = 0; z < K; z++) {
      const unsigned k = g * K + (unsigned)z;
      const bool ok = col < sxy && k < nk;
      if (ok) grid[(uint64_t)k * FLAT_PITCH(sx) * sy + (uint64_t)j * FLAT_PITCH(sx) + i] = d[z];
      // the two things the marching pass wants to know about most corners, one bit each (see flat_cut_scan_kernel):
      // word (t * 4 + wave) of plane k, bit = lane, i.e. bit (i + sx * j) of the plane in the UNPADDED corner order
      const unsigned long long ng = __ballot(ok && d[z] < 0.f), nr = __ballot(ok && dm::absf(d[z]) <= cubeDiag);
      if (k < nk && (threadIdx.x & 63u) == 0u) {
        const uint64_t wi = (uint64_t)k * (tpg * (BLOCK / 64)) + (uint64_t)t * (BLOCK / 64) + (threadIdx.x >> 6);
        negbits[wi] = ng;
        nearbits[wi] = nr;
      }
    }
  }
}

#ifndef GSDF_SPECIALIZED
// flat_march_kernel (GSDF_HIP_FLAT_STREAM=1; the default marching pass is flat_cut_scan_kernel + flat_march_list_kernel
// below): marching cubes of every cube of the lattice from the distance grid (FlatRenderer.ReadTriangles,
// glrender/flatrenderer.go:186-256). HBM-bound by design: 4 B per lattice corner in, 36 B per triangle out. Round 2:
// every WAVE on its own -- no workgroup barrier and no shared stage in the loop (round 1's kernel had one barrier per
// pass and its waves waited 68 % of their cycles at 32 % of the HBM peak).
//   * a wave pass = FLAT_TX consecutive x cubes x FLAT_ROWS rows at one z: corner 0 of the rows per lane, prefetched two
//     passes ahead; the other corners only in waves where some lane passes the reference's |d0| <= 2*sqrt3*res test
//     (:207-209) -- ~85 % of the passes end there -- and then from the neighbouring lanes / rows and one batch of ten loads;
//   * cubes the surface cuts are appended (ballot rank) as records -- 8 distances, cube coordinates, case index -- to a
//     buffer of FLAT_WAVE_RECS records in LDS that only this wave touches;
//   * when the buffer cannot take another row (> FLAT_WAVE_RECS - 64 records) the wave marches it: triangle counts per
//     record from the LDS table, exclusive prefix from three ballots, ONE global atomic for the whole flush (~350
//     triangles: ~20 K atomics per mesh, well under the ~88 per microsecond a counter word takes), then every lane builds
//     its record's triangles and stores them at their final address.
// LDS: [tri table 4 KB (row byte 15 = triangle count) | 4 x FLAT_WAVE_RECS x 10 words | 4 x 5 FLAT_WAVE_RECS u16 owner lists].
#define FLAT_ROWS 8        // cube rows of a pass
#define FLAT_TX 63         // cube columns of a pass (lane 63 supplies the last x + 1 neighbour)
#define FLAT_WAVE_RECS 160  // records per wave buffer (8 KB per wave with the owner list: four workgroups per CU)
// marching cubes of the buffered records, wave-local and balanced: an owner list (triangle -> record, number) from the
// ballot prefix sums, then ONE OUTPUT VERTEX PER LANE -- every lane busy, a wave store = 768 contiguous bytes (the first
// version built each record's triangles in its own lane: 2.2 triangles on average, 5 at most, three divisions each, and
// 36-byte pieces scattered per lane: that, not memory, was most of the 0.39 ms the active passes cost)
// KNOWN: the first triangle's index is given (`known`, wave-uniform) instead of taken from the append counter.
// Returns the number of triangles of the records.
template <int RECS, bool KNOWN = false>
__device__ __forceinline__ unsigned flat_flush_wave(uint32_t* buf, uint16_t* own, const int8_t* s_tri, unsigned& cnt, unsigned lane, float ox, float oy,
                                                    float oz, float res, float* __restrict__ tris, uint64_t tri_cap, MeshCounters* __restrict__ ctr,
                                                    unsigned long long known = 0ull) {
  auto below = [](unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };
  if (cnt == 0) return 0u;
  unsigned total = 0;
  for (unsigned i0 = 0; i0 < cnt; i0 += 64u) {
    const unsigned i = i0 + lane;
    unsigned nt = 0, idx = 0;
    if (i < cnt) { idx = buf[9 * RECS + i] >> 16; nt = (unsigned)(uint8_t)s_tri[idx * 16 + 15]; }
    const unsigned long long q0 = __ballot((nt & 1u) != 0u), q1 = __ballot((nt & 2u) != 0u), q2 = __ballot((nt & 4u) != 0u);
    const unsigned first = total + below(q0) + 2u * below(q1) + 4u * below(q2);
    for (unsigned k = 0; k < nt; k++) own[first + k] = (uint16_t)(i | (k << 8));
    total += (unsigned)__builtin_popcountll(q0) + 2u * (unsigned)__builtin_popcountll(q1) + 4u * (unsigned)__builtin_popcountll(q2);
  }
  unsigned long long gbase = known;
  if (!KNOWN) {
    if (lane == 0) gbase = atomicAdd(&ctr->n_tris, (unsigned long long)total);
    gbase = uniform_u64(gbase);
  }
  if (gbase + total > tri_cap) {  // wave-uniform: the counter keeps counting, the host learns the exact size and reruns
    if (lane == 0) ctr->overflow = 1ull;
    cnt = 0;
    return total;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the owner list is read by other lanes of this wave)
  __builtin_amdgcn_wave_barrier();
  struct __attribute__((packed, aligned(4))) V3 { float x, y, z; };
  V3* dst = (V3*)(tris + gbase * 9);
  const unsigned n3 = total * 3u;
  for (unsigned v = lane; v < n3; v += 64u) {
    const unsigned t = v / 3u, j = v - 3u * t;
    const unsigned o = own[t], i = o & 255u, k = o >> 8;
    const uint32_t xy = buf[8 * RECS + i], zi = buf[9 * RECS + i];
    const int e = s_tri[(zi >> 16) * 16 + 3u * k + (2u - j)];  // reversed winding (marchcubes.go:64-68)
    const unsigned ca = GSDF_MC_PAIR_A(e), cb = GSDF_MC_PAIR_B(e);
    // cube origin exactly as the fused round-1 kernel formed it: o + (float)index * res
    const float x0 = ox + (float)(xy & 0xffffu) * res, y0 = oy + (float)(xy >> 16) * res, z0 = oz + (float)(zi & 0xffffu) * res;
    const float x1 = x0 + res, y1 = y0 + res, z1 = z0 + res;
    const bool ax = ((ca ^ (ca >> 1)) & 1u) != 0u, ay = ((ca >> 1) & 1u) != 0u, az = ((ca >> 2) & 1u) != 0u;
    const bool bx = ((cb ^ (cb >> 1)) & 1u) != 0u, by = ((cb >> 1) & 1u) != 0u, bz = ((cb >> 2) & 1u) != 0u;
    V3 r;
    mc_interp(ax ? x1 : x0, ay ? y1 : y0, az ? z1 : z0, bx ? x1 : x0, by ? y1 : y0, bz ? z1 : z0,
              __uint_as_float(buf[ca * RECS + i]), __uint_as_float(buf[cb * RECS + i]), r.x, r.y, r.z);
#ifdef GSDF_EXP_FLAT_NO_STORE  // developer experiment: no output stream (timing only)
    if (r.x == 1.2345678e-30f) dst[v] = r;
#else
    dst[v] = r;
#endif
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // the buffer is read out before new records overwrite it
  __builtin_amdgcn_wave_barrier();
  cnt = 0;
  return total;
}

__global__ void __launch_bounds__(BLOCK, 4) flat_march_kernel(const float* __restrict__ grid, unsigned nx, unsigned ny, unsigned ncz,
                                                           unsigned czfirst, float ox, float oy, float oz, float res,
                                                           float* __restrict__ tris, uint64_t tri_cap, MeshCounters* __restrict__ ctr) {
  int8_t* s_tri = (int8_t*)g_smem;
  uint32_t* buf = (uint32_t*)(s_tri + 256 * 16) + (threadIdx.x >> 6) * (FLAT_WAVE_RECS * REC_WORDS);  // [REC_WORDS][FLAT_WAVE_RECS], this wave's
  uint16_t* own = (uint16_t*)((uint32_t*)(s_tri + 256 * 16) + 4 * FLAT_WAVE_RECS * REC_WORDS) + (threadIdx.x >> 6) * (5 * FLAT_WAVE_RECS);  // this wave's owner list
  for (int k = threadIdx.x; k < 256 * 16; k += BLOCK) s_tri[k] = (k & 15) == 15 ? (int8_t)GSDF_MC_NTRI[k >> 4] : GSDF_MC_TRI[k >> 4][k & 15];
  __syncthreads();
  const unsigned sx = FLAT_PITCH(nx + 1);  // row pitch
  const uint64_t sxy = (uint64_t)sx * (ny + 1);
  const unsigned lane = threadIdx.x & 63u;
  const unsigned txn = (nx + FLAT_TX - 1) / FLAT_TX, tyn = (ny + FLAT_ROWS - 1) / FLAT_ROWS;
  const unsigned npass = txn * tyn * ncz;  // wave passes, < 2^32 (host checks)
  const float cubeDiag = 2 * 1.73205080757f * res;  // marchcubes.go:19 / flatrenderer.go:207
  unsigned my_active = 0;  // wave-uniform
  unsigned cnt = 0;        // records in this wave's buffer (wave-uniform)
  auto below = [](unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };

  auto flush = [&]() { (void)flat_flush_wave<FLAT_WAVE_RECS>(buf, own, s_tri, cnt, lane, ox, oy, oz, res, tris, tri_cap, ctr); };

  auto pass_coords = [&](unsigned w, unsigned& tx, unsigned& ty, unsigned& cz) {
    const unsigned wr = w / txn;
    tx = w - wr * txn;
    cz = wr / tyn;
    ty = wr - cz * tyn;
  };
  // (lane 63 of a tile holds the x + 1 neighbours of lane 62's cubes and owns none itself: tiles are FLAT_TX = 63 cubes wide)
  auto load_d0 = [&](unsigned tx, unsigned ty, unsigned cz, float (&d0)[FLAT_ROWS]) {
    const unsigned cx = tx * FLAT_TX + lane, cy0 = ty * FLAT_ROWS;
    const float* g0 = grid + (uint64_t)cz * sxy + (uint64_t)cy0 * sx + cx;
#pragma unroll
    for (int r = 0; r < FLAT_ROWS; r++) d0[r] = (cx <= nx && cy0 + (unsigned)r <= ny) ? g0[(uint64_t)r * sx] : __builtin_inff();
  };
  const unsigned wid = blockIdx.x * 4u + (threadIdx.x >> 6), wstride = gridDim.x * 4u;
  // corner 0 of the next TWO passes is in flight while a pass is examined: with one (round 2's first version) a wave had
  // 2 KB outstanding, 12 waves per CU (166 registers: the seven other corners of all eight rows were loaded together) --
  // a quarter of what 8 TB/s needs at ~2 us per trip. Now 4 KB per wave at 16 waves per CU.
  float dn1[FLAT_ROWS], dn2[FLAT_ROWS];
  unsigned t1x = 0, t1y = 0, t1z = 0, t2x = 0, t2y = 0, t2z = 0;
#pragma unroll
  for (int r = 0; r < FLAT_ROWS; r++) dn1[r] = dn2[r] = __builtin_inff();
  if (wid < npass) {
    pass_coords(wid, t1x, t1y, t1z);
    load_d0(t1x, t1y, t1z, dn1);
  }
  if ((uint64_t)wid + wstride < npass) {
    pass_coords(wid + wstride, t2x, t2y, t2z);
    load_d0(t2x, t2y, t2z, dn2);
  }
  for (unsigned w = wid; w < npass; w += wstride) {  // wave-uniform
    const unsigned tx = t1x, ty = t1y, cz = t1z;
    const unsigned cx = tx * FLAT_TX + lane, cy0 = ty * FLAT_ROWS;
    const bool cube_x = lane < FLAT_TX && cx < nx;  // this lane owns a cube column
    const float* g0 = grid + (uint64_t)cz * sxy + (uint64_t)cy0 * sx + cx;
    float d0[FLAT_ROWS];
    bool any_act = false;
#pragma unroll
    for (int r = 0; r < FLAT_ROWS; r++) {
      d0[r] = dn1[r];
      dn1[r] = dn2[r];
      any_act = any_act || (cube_x && cy0 + (unsigned)r < ny && dm::absf(d0[r]) <= cubeDiag);
    }
    t1x = t2x; t1y = t2y; t1z = t2z;
    if ((uint64_t)w + 2ull * wstride < npass) {
      pass_coords(w + 2u * wstride, t2x, t2y, t2z);
      load_d0(t2x, t2y, t2z, dn2);
    }
    if (__ballot(any_act) == 0ull) continue;  // wave-uniform
#ifdef GSDF_EXP_FLAT_STREAM_ONLY  // developer experiment: the streaming read of corner 0 alone (timing only)
    my_active += 1u;
    continue;
#endif
    // The other seven corners of the active cubes. Three of them are in registers already -- x + 1 is the next lane, y + 1
    // the lane's next row -- and the plane above is fetched as nine rows at once: ONE trip to memory per active pass (the
    // first version fetched seven corners per active row, two rows at a time: four trips in a row, 0.27 ms of the
    // kernel's 0.74).
    const bool corner_x = cx <= nx;
    float up[FLAT_ROWS + 1];  // plane z + 1, rows 0..8
#pragma unroll
    for (int r = 0; r <= FLAT_ROWS; r++) up[r] = (corner_x && cy0 + (unsigned)r <= ny) ? g0[sxy + (uint64_t)r * sx] : __builtin_inff();
    const float d8 = (corner_x && cy0 + FLAT_ROWS <= ny) ? g0[(uint64_t)FLAT_ROWS * sx] : __builtin_inff();  // plane z, row 8
#pragma unroll
    for (int r = 0; r < FLAT_ROWS; r++) {
      const bool act = cube_x && cy0 + (unsigned)r < ny && dm::absf(d0[r]) <= cubeDiag;  // the reference's |d0| <= 2*sqrt3*res test (:207-209)
      const unsigned long long am = __ballot(act);
      if (am == 0ull) continue;  // wave-uniform
      my_active += (unsigned)__builtin_popcountll(am);
      const float c3 = r + 1 < FLAT_ROWS ? d0[r + 1 < FLAT_ROWS ? r + 1 : 0] : d8;
      const float c1 = __shfl_down(d0[r], 1, 64), c2 = __shfl_down(c3, 1, 64);
      const float c4 = up[r], c7 = up[r + 1], c5 = __shfl_down(c4, 1, 64), c6 = __shfl_down(c7, 1, 64);
      unsigned ix = 0;
      if (act) {
        ix = (d0[r] < 0.f ? 1u : 0u) | (c1 < 0.f ? 2u : 0u) | (c2 < 0.f ? 4u : 0u) | (c3 < 0.f ? 8u : 0u) | (c4 < 0.f ? 16u : 0u) |
             (c5 < 0.f ? 32u : 0u) | (c6 < 0.f ? 64u : 0u) | (c7 < 0.f ? 128u : 0u);
        if (ix == 255u) ix = 0u;
      }
      const unsigned long long cm = __ballot(ix != 0u);
      if (cm == 0ull) continue;  // wave-uniform
      if (cnt + 64u > FLAT_WAVE_RECS) flush();  // room for a whole row
      if (ix) {
        const unsigned pos = cnt + below(cm);
        buf[0 * FLAT_WAVE_RECS + pos] = __float_as_uint(d0[r]);
        buf[1 * FLAT_WAVE_RECS + pos] = __float_as_uint(c1);
        buf[2 * FLAT_WAVE_RECS + pos] = __float_as_uint(c2);
        buf[3 * FLAT_WAVE_RECS + pos] = __float_as_uint(c3);
        buf[4 * FLAT_WAVE_RECS + pos] = __float_as_uint(c4);
        buf[5 * FLAT_WAVE_RECS + pos] = __float_as_uint(c5);
        buf[6 * FLAT_WAVE_RECS + pos] = __float_as_uint(c6);
        buf[7 * FLAT_WAVE_RECS + pos] = __float_as_uint(c7);
        buf[8 * FLAT_WAVE_RECS + pos] = cx | ((cy0 + (unsigned)r) << 16);
        buf[9 * FLAT_WAVE_RECS + pos] = (czfirst + cz) | (ix << 16);
      }
      cnt += (unsigned)__builtin_popcountll(cm);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
  }
  flush();
  // statistics: one atomic per workgroup
  __syncthreads();
  unsigned* s_stat = (unsigned*)(s_tri + 256 * 16);
  if (lane == 0) s_stat[threadIdx.x >> 6] = my_active;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long a = (unsigned long long)s_stat[0] + s_stat[1] + s_stat[2] + s_stat[3];
    if (a) atomicAdd(&ctr->n_active, a);
  }
}

// The marching pass driven by the two bit planes flat_grid_kernel leaves beside the grid -- "distance < 0" and
// "|distance| <= 2*sqrt3*res" per lattice corner, bit (i + sx*j) of plane k in words of 64. Which cubes the surface cuts (the
// reference's corner-0 test, then a case index other than 0 and 255: flatrenderer.go:207-225, marchcubes.go:20-40), and with
// which case index, is a matter of eight shifted copies of the sign words: one LANE decides 64 cubes with a few dozen integer
// instructions and 72 bytes of loads, and only for the cubes that are cut (0.7 % at npt-flange resdiv 1600) are the eight
// distances fetched from the grid -- 1/32 of the bytes flat_march_kernel streams for the same decisions on the same
// comparisons. Two kernels, like the octree's leaf phase, because cut cubes come in slabs (a machined face parallel to a
// lattice plane cuts every cube of it) and because a counter word takes only ~88 returning atomics per microsecond:
//   flat_cut_scan_kernel   a wave pass = 64 words = 4096 consecutive cubes (corner order, rows run on) of one plane. A
//                          workgroup walks its passes twice: once counting cut cubes (a popcount), then -- after ONE atomic
//                          for the whole workgroup has reserved its stretch of the list -- again (the words are in L2),
//                          writing one 8-byte entry (case index, plane, bit) per cut cube;
//   flat_march_list_kernel the list in equal shares. A wave first adds up the triangles of its entries (count table in LDS)
//                          and reserves them with ONE atomic, then takes 64 cut cubes per trip to memory, one output vertex
//                          per lane, its triangles back to back: no atomic, no barrier in the loop.
// (Measured on the way, npt-flange resdiv 1600 / 400, where flat_march_kernel takes 0.63 / 0.08 ms: one kernel doing both,
// one atomic per flush: 0.56 / 0.18 ms -- a wave that draws a pass inside a face has 4096 cubes to march, 64 dependent trips,
// while its neighbours have none; scan + list kernels with one atomic per pass and per flush: 1.22 / 0.08 ms -- over 100 K atomics
// on two words inside 0.1 ms of work; triangle offsets fixed by the scan, three per-lane loops with a table lookup per cut
// bit: 0.33 / 0.05 ms, the scan VALU-bound at 400 instructions per pass and walk.)
#define FLATB_RECS 64    // records per wave buffer of flat_march_list_kernel: one per lane
#define FLATB_FLAGS 2048  // passes of a wave whose "holds surface" bit flat_cut_scan_kernel keeps between its two walks
#define FLATB_LDS_BYTES ((size_t)256 * 16 + (size_t)4 * FLATB_RECS * REC_WORDS * 4 + (size_t)4 * 5 * FLATB_RECS * 2)
// list entry: case index << 48 | plane (of the slab) << 32 | bit of corner 0 (unpadded corner order)
__global__ void __launch_bounds__(BLOCK) flat_cut_scan_kernel(const unsigned long long* __restrict__ negbits, const unsigned long long* __restrict__ nearbits,
                                                              unsigned wpp, unsigned nx, unsigned ny, unsigned ncz,
                                                              unsigned long long* __restrict__ list, uint64_t list_cap, MeshCounters* __restrict__ ctr) {
  __shared__ unsigned long long s_red[2][BLOCK / 64], s_base;
  __shared__ unsigned s_flag[BLOCK / 64][FLATB_FLAGS / 32];
  const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
  const unsigned sx = nx + 1;                      // corners per row: the bit planes' row length
  const uint64_t ncube_bits = (uint64_t)sx * ny;  // corner 0 of every cube has j < ny; < 2^32 (host checks sx * sy)
  const unsigned nchunk = (unsigned)((ncube_bits + 4095u) >> 12);  // passes per plane
  const unsigned wq = sx >> 6, wo = sx & 63u;  // a row further on = wq words and wo bits
  const unsigned lane_i = (lane * 64u) % sx;
  // bits [s, s + 64) of the 128-bit value hi:lo, 0 <= s <= 64
  auto funnel = [](unsigned long long lo, unsigned long long hi, unsigned s) {
    return s == 0u ? lo : (s >= 64u ? hi : ((lo >> s) | (hi << (64u - s))));
  };
  // A pass = chunk c of plane cz (both wave-uniform, advanced without divisions: the walk is (cz, c) += (dq, dr) with carry).
  // The nine words a lane needs ...
  struct Raw { unsigned long long a0, a1, a2, a3, b0, b1, b2, b3, nr; };
  auto fetch = [&](unsigned cz, unsigned c, Raw& r) {
    const unsigned wi = c * 64u + lane;  // this lane's word: cubes [64 wi, 64 wi + 64) of the plane
    const unsigned long long* n0 = negbits + (uint64_t)cz * wpp;
    const unsigned long long* n1 = n0 + wpp;  // the plane above
    // (clamped index + select, not a branch around the load: a branch makes the nine loads nine dependent trips)
    auto word = [&](const unsigned long long* pl, unsigned i) {
      const unsigned long long v = pl[i < wpp ? i : wpp - 1u];
      return i < wpp ? v : 0ull;
    };
    r.a0 = word(n0, wi); r.a1 = word(n0, wi + 1u); r.a2 = word(n0, wi + wq); r.a3 = word(n0, wi + wq + 1u);
    r.b0 = word(n1, wi); r.b1 = word(n1, wi + 1u); r.b2 = word(n1, wi + wq); r.b3 = word(n1, wi + wq + 1u);
    r.nr = word(nearbits + (uint64_t)cz * wpp, wi);
  };
  // ... and what they say: near = cubes passing the corner-0 test, cut = cubes to march, sg[c] = sign word of corner c of
  // the lane's 64 cubes (SIGNS: the second walk needs them for the case indices)
  auto decide = [&](unsigned c, const Raw& r, unsigned long long (&sg)[8], unsigned long long& near, unsigned long long& cut) {
    // the cubes of this word that exist: i < nx (a row's last corner starts no cube), j < ny
    const uint64_t bit0 = (uint64_t)(c * 64u + lane) << 6;
    unsigned long long valid = 0ull;
    if (bit0 < ncube_bits) {
      valid = ncube_bits - bit0 >= 64u ? ~0ull : ((1ull << (unsigned)(ncube_bits - bit0)) - 1ull);
      unsigned i0 = (c * 4096u) % sx + lane_i;  // x index of the word's first corner (the first term is wave-uniform)
      i0 = i0 >= sx ? i0 - sx : i0;
      for (unsigned t = i0 <= nx ? nx - i0 : nx + sx - i0; t < 64u; t += sx) valid &= ~(1ull << t);
    }
    near = r.nr & valid;
    // corner signs of the 64 cubes: x + 1 = one bit on, y + 1 = one row on (corner numbering of marchcubes.go)
    sg[0] = r.a0; sg[1] = funnel(r.a0, r.a1, 1u); sg[3] = funnel(r.a2, r.a3, wo); sg[2] = funnel(r.a2, r.a3, wo + 1u);
    sg[4] = r.b0; sg[5] = funnel(r.b0, r.b1, 1u); sg[7] = funnel(r.b2, r.b3, wo); sg[6] = funnel(r.b2, r.b3, wo + 1u);
    const unsigned long long any = sg[0] | sg[1] | sg[2] | sg[3] | sg[4] | sg[5] | sg[6] | sg[7];
    const unsigned long long all = sg[0] & sg[1] & sg[2] & sg[3] & sg[4] & sg[5] & sg[6] & sg[7];
    cut = near & any & ~all;
  };
  // case index of cube t of the lane's word: bit t of the eight sign words (32-bit selects and bit-field extracts: a 64-bit
  // shift by a per-lane amount costs four times as much)
  auto case_of = [&](const unsigned long long (&sg)[8], unsigned t) {
    const bool hi = t >= 32u;
    unsigned ix = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) ix |= (((hi ? (unsigned)(sg[k] >> 32) : (unsigned)sg[k]) >> (t & 31u)) & 1u) << k;
    return ix;
  };
  auto wave_sum = [](unsigned long long v) {
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
  };
  const uint64_t w0 = (uint64_t)blockIdx.x * (BLOCK / 64) + wv, wstep = (uint64_t)gridDim.x * (BLOCK / 64);
  const unsigned dq = (unsigned)(wstep / nchunk), dr = (unsigned)(wstep % nchunk);
  const uint64_t z0 = w0 / nchunk;  // (the only 64-bit divisions of the kernel)
  auto advance = [&](unsigned& cz, unsigned& c) {
    c += dr; cz += dq;
    if (c >= nchunk) { c -= nchunk; cz++; }
  };
  // first walk: how many. The next pass's words are in flight while a pass is decided; passes that hold surface are
  // remembered (one bit each, FLATB_FLAGS per wave; later ones are simply looked at again) so that the second walk skips the rest.
  unsigned long long my_active = 0, my_cut = 0;
  unsigned flagw = 0u, it = 0;  // wave-uniform
  if (z0 < ncz) {
    unsigned cz = (unsigned)z0, c = (unsigned)(w0 - z0 * nchunk);
    Raw cur{}, nxt{};
    fetch(cz, c, cur);
    for (; cz < ncz; it++) {  // wave-uniform
      unsigned cz2 = cz, c2 = c;
      advance(cz2, c2);
      if (cz2 < ncz) fetch(cz2, c2, nxt);
      unsigned long long sg[8], near, cut;
      decide(c, cur, sg, near, cut);
      cur = nxt; cz = cz2; c = c2;
      my_active += (unsigned long long)__builtin_popcountll(near);
      my_cut += (unsigned long long)__builtin_popcountll(cut);
      if (it < FLATB_FLAGS) {
        if (__ballot(cut != 0ull) != 0ull) flagw |= 1u << (it & 31u);
        if ((it & 31u) == 31u) { if (lane == 0) s_flag[wv][it >> 5] = flagw; flagw = 0u; }
      }
    }
    if (it < FLATB_FLAGS && (it & 31u) != 0u && lane == 0) s_flag[wv][it >> 5] = flagw;
  }
  my_active = wave_sum(my_active); my_cut = wave_sum(my_cut);
  if (lane == 0) { s_red[0][wv] = my_active; s_red[1][wv] = my_cut; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long a = 0, c = 0;
    for (int k = 0; k < BLOCK / 64; k++) { a += s_red[0][k]; c += s_red[1][k]; }
    if (a) atomicAdd(&ctr->n_active, a);
    s_base = c ? atomicAdd(&ctr->n_cut, c) : 0ull;  // the workgroup's stretch of the list (the counter keeps counting on overflow)
    if (c && s_base + c > list_cap) ctr->overflow = 1ull;
  }
  __syncthreads();
  unsigned long long at_c = s_base;  // wave-uniform: where this wave's next pass goes
  for (unsigned k = 0; k < wv; k++) at_c += s_red[1][k];
  at_c = uniform_u64(at_c);
  if (my_cut == 0ull) return;  // (wave-uniform; no barrier follows)
  // second walk: the list
  unsigned cz = (unsigned)z0, c = (unsigned)(w0 - z0 * nchunk);
  for (it = 0; cz < ncz; it++, advance(cz, c)) {  // wave-uniform
    if (it < FLATB_FLAGS && ((s_flag[wv][it >> 5] >> (it & 31u)) & 1u) == 0u) continue;  // (wave-uniform) known to hold no surface
    Raw cur;
    unsigned long long sg[8], near, cut;
    fetch(cz, c, cur);
    decide(c, cur, sg, near, cut);
    if (__ballot(cut != 0ull) == 0ull) continue;  // wave-uniform: no surface in these 4096 cubes
    const unsigned pc = (unsigned)__builtin_popcountll(cut);
    unsigned pre = pc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = __shfl_up(pre, d, 64);
      if (lane >= (unsigned)d) pre += o;
    }
    const unsigned tot = __shfl(pre, 63, 64);  // wave-uniform
    unsigned long long ec = at_c + (pre - pc);
    const unsigned long long where0 = ((unsigned long long)cz << 32) | ((uint64_t)(c * 64u + lane) << 6);
    for (unsigned long long m = cut; m != 0ull; m &= m - 1ull, ec++) {  // (as long as the wave's fullest lane)
      const unsigned t = (unsigned)__builtin_ctzll(m);
      if (ec < list_cap) list[ec] = ((unsigned long long)case_of(sg, t) << 48) | (where0 + t);
    }
    at_c += tot;
  }
}

// LDS: [tri table 4 KB | 4 x FLATB_RECS x 10 words | 4 x 5 FLATB_RECS u16 owner lists] = 16.5 KB; 80 registers: six workgroups per CU (at eight the kernel spills).
__global__ void __launch_bounds__(BLOCK, 6) flat_march_list_kernel(const float* __restrict__ grid, const unsigned long long* __restrict__ list,
                                                                uint64_t list_cap, unsigned nx, unsigned ny, unsigned czfirst, float ox, float oy,
                                                                float oz, float res, float* __restrict__ tris, uint64_t tri_cap,
                                                                MeshCounters* __restrict__ ctr) {
  int8_t* s_tri = (int8_t*)g_smem;
  const unsigned wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  uint32_t* buf = (uint32_t*)(s_tri + 256 * 16) + wv * (FLATB_RECS * REC_WORDS);  // [REC_WORDS][FLATB_RECS], this wave's
  uint16_t* own = (uint16_t*)((uint32_t*)(s_tri + 256 * 16) + 4 * FLATB_RECS * REC_WORDS) + wv * (5 * FLATB_RECS);
  unsigned long long n = uniform_u64(ctr->n_cut);  // written by flat_cut_scan_kernel, the launch before
  for (int k = threadIdx.x; k < 256 * 16; k += BLOCK) s_tri[k] = (k & 15) == 15 ? (int8_t)GSDF_MC_NTRI[k >> 4] : GSDF_MC_TRI[k >> 4][k & 15];
  __syncthreads();
  if (n > list_cap) n = list_cap;  // (the host reruns both kernels with room; the triangles are still counted)
  const unsigned sx = nx + 1;
  const unsigned pitch = FLAT_PITCH(sx);  // the grid's row pitch
  const uint64_t pxy = (uint64_t)pitch * (ny + 1);
  // equal shares of whole 64-entry chunks, contiguous per wave (neighbouring entries are neighbouring cubes: their corner
  // lines are shared)
  const uint64_t nchunks = (n + 63u) >> 6, nwaves = (uint64_t)gridDim.x * 4u, me = (uint64_t)blockIdx.x * 4u + wv;
  const uint64_t c_lo = nchunks * me / nwaves, c_hi = nchunks * (me + 1u) / nwaves;
  auto entry = [&](uint64_t ch) {
    const uint64_t at = (ch << 6) + lane;
    return list[at < n ? at : n - 1u];  // (padding lanes of the last chunk read a valid entry; nothing of it is kept)
  };
  // how many triangles this workgroup's entries make, and where they go: one atomic per workgroup (one per wave -- 8 000 of
  // them on one word as the kernel starts -- cost 60 us)
  __shared__ unsigned long long s_mine[4], s_base;
  unsigned long long mine = 0;
  for (uint64_t ch = c_lo; ch < c_hi; ch++) {
    const unsigned long long e = entry(ch);
    if ((ch << 6) + lane < n) mine += (unsigned long long)(uint8_t)s_tri[(unsigned)(e >> 48) * 16u + 15u];
  }
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
  if (lane == 0) s_mine[wv] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long all = s_mine[0] + s_mine[1] + s_mine[2] + s_mine[3];
    s_base = all ? atomicAdd(&ctr->n_tris, all) : 0ull;
    if (all && s_base + all > tri_cap) ctr->overflow = 1ull;  // the counter keeps counting: the host learns the exact size and reruns
  }
  __syncthreads();
  if (c_lo >= c_hi) return;  // (wave-uniform; no barrier follows)
  unsigned long long base = s_base;
  for (unsigned k = 0; k < wv; k++) base += s_mine[k];
  base = uniform_u64(base);
  // 64 cut cubes per trip: a chunk = one record per lane = one flush; the next chunk's distances and the list entries of the
  // one after are in flight meanwhile
  struct Got { float d[8]; unsigned ci, cj, cz, ix; };
  auto gather = [&](unsigned long long e, Got& g) {
    const unsigned bit = (unsigned)e;
    g.ix = (unsigned)(e >> 48);
    g.cz = (unsigned)(e >> 32) & 0xffffu;
    g.cj = bit / sx;
    g.ci = bit - g.cj * sx;
    const float* g0 = grid + (uint64_t)g.cz * pxy + (uint64_t)g.cj * pitch + g.ci;
    g.d[0] = g0[0]; g.d[1] = g0[1]; g.d[3] = g0[pitch]; g.d[2] = g0[pitch + 1u];
    g.d[4] = g0[pxy]; g.d[5] = g0[pxy + 1u]; g.d[7] = g0[pxy + pitch]; g.d[6] = g0[pxy + pitch + 1u];
  };
  unsigned long long e_nxt = entry(c_lo);
  Got cur{}, nxt{};
  gather(e_nxt, cur);
  if (c_lo + 1u < c_hi) e_nxt = entry(c_lo + 1u);
  unsigned cnt = 0;
  for (uint64_t ch = c_lo; ch < c_hi; ch++) {  // wave-uniform
    if (ch + 1u < c_hi) gather(e_nxt, nxt);
    if (ch + 2u < c_hi) e_nxt = entry(ch + 2u);
    const unsigned here = n - (ch << 6) < 64u ? (unsigned)(n - (ch << 6)) : 64u;  // records of this chunk (wave-uniform)
    if (lane < here) {
#pragma unroll
      for (int k = 0; k < 8; k++) buf[k * FLATB_RECS + lane] = __float_as_uint(cur.d[k]);
      buf[8 * FLATB_RECS + lane] = cur.ci | (cur.cj << 16);
      buf[9 * FLATB_RECS + lane] = (czfirst + cur.cz) | (cur.ix << 16);  // (the case index the sign bits gave: the same comparisons on the same values)
    }
    cnt = here;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    base += flat_flush_wave<FLATB_RECS, true>(buf, own, s_tri, cnt, lane, ox, oy, oz, res, tris, tri_cap, ctr, base);
    cur = nxt;
  }
}
#endif  // GSDF_SPECIALIZED

// =================================================================================================
// Dual contouring on device (glrender/dual_contour.go, dual_contour_vertexplacement.go).
// The reference keeps a map[i3.Vec]int over a full BFS decomposition; here the lattice is a dense
// int32 index grid in HBM (levels <= 11 -> <= 4.3 GB, trivial against 288 GB), so neighbour lookups
// are single loads and every stage is one lane per cube / per active edge.
// =================================================================================================
struct DCCounters {
  unsigned long long n_cubes, n_edges, n_tris, q_overflow, t_overflow;
  unsigned long long n_origin_evals;  // lattice cells whose origin was actually evaluated (the rest were outside the exact box)
};

// Stage 1 (Reset :26-83): evaluate every cube origin; keep iff |d| < 2*size (octreePrunea szMult=2, origin).
template <int K, int W = 3>
__global__ void __launch_bounds__(BLOCK, W) dc_origin_kernel(const uint32_t* __restrict__ code_g, int nslots, int nshift, float ox, float oy,
                                                             float oz, float res, int* __restrict__ grid, Cube* __restrict__ cubes,
                                                             unsigned long long cube_cap, unsigned zlo, unsigned zhi,
                                                             int use_box, float bx0, float by0, float bz0, float bx1, float by1,
                                                             float bz1, unsigned tx0, unsigned ty0, unsigned tz0, unsigned ntx,
                                                             unsigned nty, unsigned ntz, DCCounters* __restrict__ ctr) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  unsigned* s_w = (unsigned*)(g_smem + (size_t)(nslots > 0 ? nslots : 1) * K * BLOCK);  // 4 wave totals
  unsigned long long* s_base = (unsigned long long*)(s_w + 4);
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // multi-GPU: this rank evaluates the z-slab [zlo, zhi) of the lattice (its own slab plus a one-cube halo).
  // Work tile = a compact 8 x 8 x 4K brick of cells (a wave = one 8x8 patch at K z-levels), not a K*256-long row:
  // spatially coherent waves are what lets D_SKIPFAR* drop the far children of a wide union for the whole wave.
  // The host hands over the range of tiles to sweep: all of the slab, or -- for trees with an exact box -- the tiles
  // that can hold a kept cube or a neighbour of one (box grown by the keep radius and two cells); cells outside that
  // range are neither written here nor read by the later stages.
  const unsigned n = 1u << nshift;
  const uint64_t ntiles = (uint64_t)ntx * nty * ntz;
  const float maxDist = res * 2;
  unsigned long long my_evals = 0;
  for (uint64_t T = blockIdx.x; T < ntiles; T += gridDim.x) {  // block-uniform trip count
    const unsigned tx = tx0 + (unsigned)(T % ntx), ty = ty0 + (unsigned)((T / ntx) % nty), tz = tz0 + (unsigned)(T / ((uint64_t)ntx * nty));
    P3 p[K];
    float d[K];
    unsigned cx[K], cy[K], cz[K];
    bool valid[K];
#pragma unroll
    for (int kp = 0; kp < K; kp++) {
      const unsigned j = (unsigned)kp * BLOCK + threadIdx.x;
      cx[kp] = tx * 8u + (j & 7u);
      cy[kp] = ty * 8u + ((j >> 3) & 7u);
      cz[kp] = zlo + tz * (4u * K) + (j >> 6);
      valid[kp] = cx[kp] < n && cy[kp] < n && cz[kp] < zhi;
      p[kp] = P3{ox + res * (float)cx[kp], oy + res * (float)cy[kp], oz + res * (float)cz[kp]};  // CubeOrigin, size = res
    }
    // Trees whose field is >= the distance to a known box (use_box): a wave whose cells all lie outside that box by
    // more than the keep radius needs no evaluation -- |d| >= distance to the box > 2*res decides "not kept" exactly.
    // (The reference sweeps its cubic lattice unconditionally; a long thin part fills a few percent of it.)
    bool far = use_box != 0;
#pragma unroll
    for (int kp = 0; kp < K; kp++) {
      const float L = dm::maxf(dm::maxf(dm::maxf(bx0 - p[kp].x, p[kp].x - bx1), dm::maxf(by0 - p[kp].y, p[kp].y - by1)),
                               dm::maxf(bz0 - p[kp].z, p[kp].z - bz1));
      far = far && (!valid[kp] || L > maxDist * 1.001f);
    }
    if (__all(far)) {
#pragma unroll
      for (int kp = 0; kp < K; kp++) d[kp] = 3.0e38f;
    } else {
      gsdf_dev::sdf_eval<K, 2>(code, p, d, lds, BLOCK);  // COLUMN mode: the lane's K cells are one x,y column (z = wave + 4 kp)
#pragma unroll
      for (int kp = 0; kp < K; kp++) {  // count the lattice cells (tiles overhang the lattice edge)
        const unsigned long long vm = __ballot(valid[kp]);
        if (lane == 0) my_evals += (unsigned long long)__builtin_popcountll(vm);
      }
    }
    // Block-wide append: ONE global atomic per workgroup pass (K*256 cells) instead of one per wave and point --
    // at ~88 atomics/us on a single word the per-wave form was a co-bottleneck for cheap trees (1e9 cells / 64).
    bool keep[K];
    unsigned mine = 0;
#pragma unroll
    for (int kp = 0; kp < K; kp++) {
      keep[kp] = valid[kp] && !(dm::absf(d[kp]) >= maxDist);
      mine += keep[kp] ? 1u : 0u;
    }
    unsigned incl = mine;  // wave inclusive scan of per-lane counts
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned v = __shfl_up(incl, off, 64);
      if (lane >= (unsigned)off) incl += v;
    }
    __syncthreads();  // previous pass finished reading s_w / s_base
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    const unsigned w0 = s_w[0], w1 = s_w[1], w2 = s_w[2], w3 = s_w[3];
    const unsigned total = w0 + w1 + w2 + w3;
    if (threadIdx.x == 0 && total) *s_base = atomicAdd(&ctr->n_cubes, (unsigned long long)total);
    __syncthreads();
    unsigned long long slot = (total ? *s_base : 0ull) + (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u) + (incl - mine);
#pragma unroll
    for (int kp = 0; kp < K; kp++) {
      const uint64_t c = (uint64_t)cx[kp] + ((uint64_t)cy[kp] << nshift) + ((uint64_t)cz[kp] << (2 * nshift));
      if (keep[kp]) {
        if (slot < cube_cap) {
          cubes[slot] = Cube{(uint16_t)cx[kp], (uint16_t)cy[kp], (uint16_t)cz[kp], 0};
          grid[c] = (int)slot;
        } else {
          ctr->q_overflow = 1ull;
          grid[c] = -1;
        }
        slot++;
      } else if (valid[kp]) {
        grid[c] = -1;
      }
    }
  }
  if (lane == 0 && my_evals) atomicAdd(&ctr->n_origin_evals, my_evals);
}

// Stage 2 (RenderAll :85-108): origin, +x, +y, +z distances of every kept cube (one 4-point pass per
// lane); default FinalVertex = cube origin; active edges (sign BIT differs, :261-269) are compacted.
__global__ void __launch_bounds__(BLOCK, 3) dc_edges_kernel(const uint32_t* __restrict__ code_g, const Cube* __restrict__ cubes,
                                                            unsigned long long cube_cap, float ox, float oy, float oz, float res,
                                                            float4* __restrict__ dists, float* __restrict__ fv,
                                                            unsigned* __restrict__ edges, unsigned long long edge_cap,
                                                            DCCounters* __restrict__ ctr) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  unsigned long long n = uniform_u64(ctr->n_cubes);
  if (n > cube_cap) n = cube_cap;
  const uint64_t step = (uint64_t)gridDim.x * BLOCK;
  for (uint64_t base = (uint64_t)blockIdx.x * BLOCK; base < n; base += step) {
    const uint64_t i = base + threadIdx.x;
    const bool valid = i < n;
    Cube c = {0, 0, 0, 0};
    if (valid) c = cubes[i];
    const float x0 = ox + res * (float)c.x, y0 = oy + res * (float)c.y, z0 = oz + res * (float)c.z;
    P3 p[4] = {{x0, y0, z0}, {x0 + res, y0 + 0.f, z0 + 0.f}, {x0 + 0.f, y0 + res, z0 + 0.f}, {x0 + 0.f, y0 + 0.f, z0 + res}};
    float d[4];
    gsdf_dev::sdf_eval<4>(code, p, d, lds, BLOCK);
    if (valid) {
      dists[i] = make_float4(d[0], d[1], d[2], d[3]);
      fv[3 * i] = x0; fv[3 * i + 1] = y0; fv[3 * i + 2] = z0;
    }
    const unsigned s0 = __float_as_uint(d[0]) >> 31;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const bool act = valid && ((__float_as_uint(d[1 + a]) >> 31) != s0);
      const unsigned long long slot = wave_append(act, &ctr->n_edges);
      if (act) {
        if (slot < edge_cap) edges[slot] = ((unsigned)i << 2) | (unsigned)a;
        else ctr->q_overflow = 1ull;
      }
    }
  }
}

__device__ __forceinline__ float dc_isect(float o, float e) { return -o / (e - o); }

// Stage 3 (PlaceVertices :28-50 + gleval.NormalsCentralDiff): raw central-difference normals at the
// linear intersection of every ACTIVE edge (inactive edges' normals are never read by the reference).
__global__ void __launch_bounds__(BLOCK, 3) dc_normals_kernel(const uint32_t* __restrict__ code_g, const Cube* __restrict__ cubes,
                                                              const float4* __restrict__ dists, const unsigned* __restrict__ edges,
                                                              unsigned long long edge_cap, float ox, float oy, float oz, float res,
                                                              float h, float* __restrict__ nrm, DCCounters* __restrict__ ctr) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  unsigned long long n = uniform_u64(ctr->n_edges);
  if (n > edge_cap) n = edge_cap;
  const uint64_t step = (uint64_t)gridDim.x * BLOCK;
  for (uint64_t base = (uint64_t)blockIdx.x * BLOCK; base < n; base += step) {
    const uint64_t i = base + threadIdx.x;
    const bool valid = i < n;
    float px = 0, py = 0, pz = 0;
    unsigned e = 0;
    if (valid) {
      e = edges[i];
      const unsigned ci = e >> 2, a = e & 3u;
      const Cube c = cubes[ci];
      const float4 d = dists[ci];
      const float t = res * dc_isect(d.x, a == 0 ? d.y : (a == 1 ? d.z : d.w));
      px = (ox + res * (float)c.x) + (a == 0 ? t : 0.f);
      py = (oy + res * (float)c.y) + (a == 1 ? t : 0.f);
      pz = (oz + res * (float)c.z) + (a == 2 ? t : 0.f);
    }
    float out[3];
#pragma unroll 1
    for (int dim = 0; dim < 3; dim++) {
      P3 ab[2] = {{px + (dim == 0 ? h : 0.f), py + (dim == 1 ? h : 0.f), pz + (dim == 2 ? h : 0.f)},
                  {px - (dim == 0 ? h : 0.f), py - (dim == 1 ? h : 0.f), pz - (dim == 2 ? h : 0.f)}};
      float dd[2];
      gsdf_dev::sdf_eval<2>(code, ab, dd, lds, BLOCK);
      const float v = dd[0] - dd[1];
      if (dim == 0) out[0] = v; else if (dim == 1) out[1] = v; else out[2] = v;
    }
    if (valid) {
      const size_t o = ((size_t)(e >> 2) * 3 + (e & 3u)) * 3;
      nrm[o] = out[0]; nrm[o + 1] = out[1]; nrm[o + 2] = out[2];
    }
  }
}

#define DC_ROWS 18  // <= 3 own + 12 contributed (own edges appear again among them) + 3 regularisation rows
#define DC_BLOCK 64
// Stage 4 (PlaceVertices :52-141, leastSquaresMGS64 :152-223): per cube, rows = own active edges, then the
// edges of the (up to 12) contributing cubes in lattice order (z,y,x) and axis order, 3 regularisation rows;
// float64 modified Gram-Schmidt with the rows staged in LDS ([row][col][lane]; zero rows are exact no-ops).
__global__ void __launch_bounds__(DC_BLOCK) dc_place_kernel(const Cube* __restrict__ cubes, unsigned long long cube_cap,
                                                            const float4* __restrict__ dists, const int* __restrict__ grid,
                                                            const float* __restrict__ nrm, int nshift, float ox, float oy, float oz,
                                                            float res, float sqrtLambda, float* __restrict__ fv, unsigned zplace_hi,
                                                            DCCounters* __restrict__ ctr) {
  __shared__ double sQ[DC_ROWS][3][DC_BLOCK];
  __shared__ double sB[DC_ROWS][DC_BLOCK];
  unsigned long long n = uniform_u64(ctr->n_cubes);
  if (n > cube_cap) n = cube_cap;
  const int nn = 1 << nshift;
  const unsigned t = threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * DC_BLOCK;
  for (uint64_t base = (uint64_t)blockIdx.x * DC_BLOCK; base < n; base += step) {
    const uint64_t i = base + t;
    if (i >= n) continue;  // no block-level sync below: each lane owns column t of the LDS arrays
    const Cube c = cubes[i];
    if (c.z >= zplace_hi) continue;  // top halo layer: only its distances/normals are needed
    const float cox = ox + res * (float)c.x, coy = oy + res * (float)c.y, coz = oz + res * (float)c.z;
    const float invRes = 1.0f / res;
    int nr = 0, nnb = 0;
    float mx = 0.f, my = 0.f, mz = 0.f;
    auto add_row = [&](float bx, float by, float bz, float nx, float ny, float nz) {
      const float qx = invRes * (bx - cox), qy = invRes * (by - coy), qz = invRes * (bz - coz);
      sQ[nr][0][t] = (double)nx; sQ[nr][1][t] = (double)ny; sQ[nr][2][t] = (double)nz;
      sB[nr][t] = (double)(nx * qx + ny * qy + nz * qz);
      mx = mx + bx; my = my + by; mz = mz + bz;
      nr++;
    };
    auto edge_row = [&](unsigned ci, int a) {
      const Cube u = cubes[ci];
      const float4 d = dists[ci];
      const float tt = res * dc_isect(d.x, a == 0 ? d.y : (a == 1 ? d.z : d.w));
      const float ux = ox + res * (float)u.x, uy = oy + res * (float)u.y, uz = oz + res * (float)u.z;
      const size_t o = ((size_t)ci * 3 + (size_t)a) * 3;
      add_row(ux + (a == 0 ? tt : 0.f), uy + (a == 1 ? tt : 0.f), uz + (a == 2 ? tt : 0.f), nrm[o], nrm[o + 1], nrm[o + 2]);
    };
    // neighbour records first decide whether this cube is placed at all (len(cube.Neighbors) == 0 -> skip)
    unsigned contrib[12];
    unsigned char caxis[12];
    for (int dz = 0; dz < 2; dz++)
      for (int dy = 0; dy < 2; dy++)
        for (int dx = 0; dx < 2; dx++) {
          const int ux = c.x + dx, uy = c.y + dy, uz = c.z + dz;
          if (ux >= nn || uy >= nn || uz >= nn) continue;
          const int ui = grid[((size_t)uz * nn + uy) * nn + ux];
          if (ui < 0) continue;
          const float4 d = dists[ui];
          const unsigned s0 = __float_as_uint(d.x) >> 31;
          if (dx == 0 && ((__float_as_uint(d.y) >> 31) != s0)) { contrib[nnb] = (unsigned)ui; caxis[nnb++] = 0; }
          if (dy == 0 && ((__float_as_uint(d.z) >> 31) != s0)) { contrib[nnb] = (unsigned)ui; caxis[nnb++] = 1; }
          if (dz == 0 && ((__float_as_uint(d.w) >> 31) != s0)) { contrib[nnb] = (unsigned)ui; caxis[nnb++] = 2; }
        }
    if (nnb == 0) continue;
    {
      const float4 d = dists[i];
      const unsigned s0 = __float_as_uint(d.x) >> 31;
      if ((__float_as_uint(d.y) >> 31) != s0) edge_row((unsigned)i, 0);
      if ((__float_as_uint(d.z) >> 31) != s0) edge_row((unsigned)i, 1);
      if ((__float_as_uint(d.w) >> 31) != s0) edge_row((unsigned)i, 2);
    }
    for (int k = 0; k < nnb; k++) edge_row(contrib[k], caxis[k]);
    const float im = 1.f / (float)nr;
    const float bsx = invRes * (im * mx - cox), bsy = invRes * (im * my - coy), bsz = invRes * (im * mz - coz);
    sQ[nr][0][t] = (double)sqrtLambda; sQ[nr][1][t] = 0.0; sQ[nr][2][t] = 0.0; sB[nr][t] = (double)(sqrtLambda * bsx); nr++;
    sQ[nr][0][t] = 0.0; sQ[nr][1][t] = (double)sqrtLambda; sQ[nr][2][t] = 0.0; sB[nr][t] = (double)(sqrtLambda * bsy); nr++;
    sQ[nr][0][t] = 0.0; sQ[nr][1][t] = 0.0; sQ[nr][2][t] = (double)sqrtLambda; sB[nr][t] = (double)(sqrtLambda * bsz); nr++;
    const int K = nr;
    double R[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int j = 0; j < 3; j++) {
      for (int ii = 0; ii < j; ii++) {
        double dot = 0;
        for (int k = 0; k < K; k++) dot += sQ[k][ii][t] * sQ[k][j][t];
        R[ii][j] = dot;
        for (int k = 0; k < K; k++) sQ[k][j][t] -= dot * sQ[k][ii][t];
      }
      double nsq = 0;
      for (int k = 0; k < K; k++) nsq += sQ[k][j][t] * sQ[k][j][t];
      const double norm = __builtin_sqrt(nsq);
      R[j][j] = norm;
      if (norm > 1e-14) {
        const double inv = 1.0 / norm;
        for (int k = 0; k < K; k++) sQ[k][j][t] *= inv;
      }
    }
    double Qtb[3] = {0, 0, 0};
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < K; k++) Qtb[j] += sQ[k][j][t] * sB[k][t];
    double x[3];
    for (int ii = 2; ii >= 0; ii--) {
      x[ii] = Qtb[ii];
      for (int k = ii + 1; k < 3; k++) x[ii] -= R[ii][k] * x[k];
      if (R[ii][ii] > 1e-14) x[ii] /= R[ii][ii];
      else x[ii] = 0;
    }
    const float xf = dm::clampf((float)x[0], -0.1f, 1.1f), yf = dm::clampf((float)x[1], -0.1f, 1.1f), zf = dm::clampf((float)x[2], -0.1f, 1.1f);
    fv[3 * i] = res * xf + cox; fv[3 * i + 1] = res * yf + coy; fv[3 * i + 2] = res * zf + coz;
  }
}

// Stage 5 (RenderAll :143-219): one quad (2 triangles) per active edge whose 4 surrounding cubes exist.
__global__ void __launch_bounds__(BLOCK) dc_quads_kernel(const Cube* __restrict__ cubes, const float4* __restrict__ dists,
                                                         const unsigned* __restrict__ edges, unsigned long long edge_cap,
                                                         const int* __restrict__ grid, const float* __restrict__ fv, int nshift,
                                                         unsigned zown_lo, unsigned zown_hi, float* __restrict__ tris,
                                                         unsigned long long tri_cap, DCCounters* __restrict__ ctr) {
  unsigned long long n = uniform_u64(ctr->n_edges);
  if (n > edge_cap) n = edge_cap;
  const int nn = 1 << nshift;
  const uint64_t step = (uint64_t)gridDim.x * BLOCK;
  for (uint64_t base = (uint64_t)blockIdx.x * BLOCK; base < n; base += step) {
    const uint64_t i = base + threadIdx.x;
    bool ok = i < n;
    int q[4] = {-1, -1, -1, -1};
    bool flip = false;
    if (ok) {
      const unsigned e = edges[i];
      const unsigned ci = e >> 2, a = e & 3u;
      const Cube c = cubes[ci];
      const float4 d = dists[ci];
      ok = ok && c.z >= zown_lo && c.z < zown_hi;  // quads are emitted by the rank that owns the edge's cube
      flip = ((a == 0 ? d.y : (a == 1 ? d.z : d.w)) - d.x) < 0.f;
      // EdgeNeighborsX/Y/Z (:271-287): offsets in cube units
      const int off[3][4][3] = {{{0, -1, -1}, {0, 0, -1}, {0, 0, 0}, {0, -1, 0}},
                                {{-1, 0, -1}, {-1, 0, 0}, {0, 0, 0}, {0, 0, -1}},
                                {{-1, -1, 0}, {0, -1, 0}, {0, 0, 0}, {-1, 0, 0}}};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int x = c.x + off[a][k][0], y = c.y + off[a][k][1], z = c.z + off[a][k][2];
        int idx = -1;
        if (x >= 0 && y >= 0 && z >= 0 && x < nn && y < nn && z < nn) idx = grid[((size_t)z * nn + y) * nn + x];
        q[k] = idx;
        ok = ok && idx >= 0;
      }
    }
    const unsigned long long slot = wave_append(ok, &ctr->n_tris);
    if (ok) {
      if (2 * slot + 2 <= tri_cap) {
        int o[4] = {q[0], q[1], q[2], q[3]};
        if (flip) { o[0] = q[3]; o[1] = q[2]; o[2] = q[1]; o[3] = q[0]; }
        float* dst = tris + 18 * slot;
        const int order[6] = {0, 1, 2, 2, 3, 0};
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const float* v = fv + 3 * (size_t)o[order[k]];
          dst[3 * k] = v[0]; dst[3 * k + 1] = v[1]; dst[3 * k + 2] = v[2];
        }
      } else {
        ctr->t_overflow = 1ull;
      }
    }
  }
}

// glrender.ImageRendererSDF2.Render (image.go:76-118) with the default black/white/red conversion (:52-61):
// pixel (i,j) samples (xmin + i*dx, ymax - j*dy); dist gets the raw distances, rgba the converted pixels.
template <int K>
__global__ void __launch_bounds__(BLOCK, 3) image2_kernel(const uint32_t* __restrict__ code_g, int w, int h, float xmin, float ymax,
                                                          float dx, float dy, float* __restrict__ dist, uint32_t* __restrict__ rgba) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  const uint64_t n = (uint64_t)w * (uint64_t)h;
  const uint64_t step = (uint64_t)gridDim.x * BLOCK * K;
  for (uint64_t base = (uint64_t)blockIdx.x * BLOCK * K; base < n; base += step) {
    P3 p[K];
    float d[K];
#pragma unroll
    for (int kp = 0; kp < K; kp++) {
      uint64_t i = base + (uint64_t)kp * BLOCK + threadIdx.x;
      if (i >= n) i = n - 1;
      const unsigned px = (unsigned)(i % (uint64_t)w), py = (unsigned)(i / (uint64_t)w);
      p[kp] = P3{(float)px * dx + xmin, ymax - (float)py * dy, 0.f};
    }
    gsdf_dev::sdf_eval<K>(code, p, d, lds, BLOCK);
#pragma unroll
    for (int kp = 0; kp < K; kp++) {
      const uint64_t i = base + (uint64_t)kp * BLOCK + threadIdx.x;
      if (i < n) {
        const float v = d[kp];
        if (dist) dist[i] = v;
        const bool bad = (v != v) || (dm::absf(v) == __builtin_inff());
        if (rgba) rgba[i] = bad ? 0xff0000ffu : (v > 0.f ? 0xffffffffu : 0xff000000u);  // R,G,B,A bytes little-endian
      }
    }
  }
}

// Exhaustive self-test of dm::sqrt_1to2 over every float in [1, 2].
__global__ void __launch_bounds__(BLOCK) sqrt_selftest_kernel(unsigned long long* __restrict__ bad) {
  unsigned long long nb = 0;
  for (unsigned i = 0x3f800000u + blockIdx.x * BLOCK + threadIdx.x; i <= 0x40000000u; i += gridDim.x * BLOCK) {
    const float s = __uint_as_float(i);
    if (__float_as_uint(dm::sqrt_1to2(s)) != __float_as_uint(__builtin_sqrtf(s))) nb++;
  }
  if (nb) atomicAdd(bad, nb);
}

// Self-test of dm::circ_sector_fast against the reference's expression floor(float32(atan2(y, x)) / angle): 2^32 points --
// even indices: both coordinates from a hash, all magnitudes and signs (exponents 2^-40 .. 2^40, plus zeros); odd indices:
// points within 1e-3 .. 1e-9 rad of a sector boundary at radii 1e-2 .. 1e2. bad counts the points where the fast path decides
// and differs; fast_count the points where it decides.
__global__ void __launch_bounds__(BLOCK) circ_selftest_kernel(float angle, unsigned long long* __restrict__ bad,
                                                              unsigned long long* __restrict__ fast_count) {
  unsigned long long nb = 0, nf = 0;
  const float inv_angle = __builtin_amdgcn_rcpf(angle), m = 6e-6f * inv_angle;
  auto hash = [](unsigned v) { v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16; return v; };
  for (unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * BLOCK) {
    const unsigned h1 = hash((unsigned)i), h2 = hash((unsigned)i ^ 0x9e3779b9u);
    float x, y;
    if ((i & 1ull) == 0ull) {
      // sign | exponent 87..167 | 23 mantissa bits; one in 64 is a zero
      x = __uint_as_float((h1 & 0x80000000u) | ((87u + (h1 >> 8) % 81u) << 23) | (h2 & 0x7fffffu));
      y = __uint_as_float((h2 & 0x80000000u) | ((87u + (h2 >> 8) % 81u) << 23) | (h1 & 0x7fffffu));
      if ((h1 & 63u) == 0u) x = (h2 & 1u) ? 0.f : -0.f;
      if ((h2 & 63u) == 1u) y = (h1 & 1u) ? 0.f : -0.f;
    } else {
      const int k = (int)(h1 % 2001u) - 1000;
      const float eps = __builtin_exp2f(-10.f - 20.f * (float)(h2 & 0xffffu) * (1.f / 65536.f)) * ((h2 & 0x10000u) ? 1.f : -1.f);
      const float th = (float)k * angle + eps, rad = __builtin_exp2f(-6.6f + 13.2f * (float)(h1 >> 16) * (1.f / 65536.f));
      float sn, cs;
      dm::sincosf_(th, sn, cs);
      x = rad * cs; y = rad * sn;
    }
    float id;
    if (dm::circ_sector_fast(x, y, inv_angle, m, id)) {
      nf++;
      const float ref = dm::floorf_(dm::atan2f_(y, x) / angle);
      if (__float_as_uint(id) != __float_as_uint(ref) && !(id == 0.f && ref == 0.f)) nb++;
    }
  }
  if (nb) atomicAdd(bad, nb);
  if (nf) atomicAdd(fast_count, nf);
}

// Exhaustive self-test of dm::div_by_uniform: every float32 numerator against the IEEE division.
__global__ void __launch_bounds__(BLOCK) div_selftest_kernel(float d, float r, unsigned long long* __restrict__ bad,
                                                             unsigned long long* __restrict__ fast_count) {
  unsigned long long nb = 0, nf = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * BLOCK) {
    const float n = __uint_as_float((unsigned)i);
    if (!dm::div_fast_ok(n)) continue;  // the interpreter takes the IEEE path for these
    nf++;
    const float a = dm::div_by_uniform(n, d, r), b = n / d;
    if (__float_as_uint(a) != __float_as_uint(b)) nb++;
  }
  if (nb) atomicAdd(bad, nb);
  atomicAdd(fast_count, nf);
}

// STL records (stl.go:15-62): one wave stages 64 x 50-byte records in LDS, then stores dwords.
__global__ void __launch_bounds__(BLOCK) stl_kernel(const float* __restrict__ tris, uint64_t n, uint8_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint8_t stage[BLOCK * 50];
  const uint64_t step = (uint64_t)gridDim.x * BLOCK;
  for (uint64_t base = (uint64_t)blockIdx.x * BLOCK; base < n; base += step) {
    const uint64_t i = base + threadIdx.x;
    if (i < n) {
      const float* t = tris + 9 * i;
      float v[9];
#pragma unroll
      for (int k = 0; k < 9; k++) v[k] = t[k];
      // Unit(Cross(t1-t0, t2-t0)) with ms3.Norm = nested hypot
      const float ax = v[3] - v[0], ay = v[4] - v[1], az = v[5] - v[2];
      const float cx = v[6] - v[0], cy = v[7] - v[1], cz = v[8] - v[2];
      const float nx = ay * cz - az * cy, ny = az * cx - ax * cz, nz = ax * cy - ay * cx;
      const float inv = 1.0f / dm::norm3(nx, ny, nz);
      uint16_t* rec = (uint16_t*)(stage + threadIdx.x * 50);
      float f[12] = {inv * nx, inv * ny, inv * nz, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]};
#pragma unroll
      for (int k = 0; k < 12; k++) {
        const uint32_t u = __float_as_uint(f[k]);
        rec[2 * k] = (uint16_t)(u & 0xffffu);
        rec[2 * k + 1] = (uint16_t)(u >> 16);
      }
      rec[24] = 0;
    }
    __syncthreads();
    const uint64_t nrec = (n - base) < BLOCK ? (n - base) : BLOCK;
    const uint64_t nbytes = nrec * 50;
    uint8_t* o = out + 84 + base * 50;  // base is a multiple of 256 -> 4-byte aligned
    const uint32_t* s32 = (const uint32_t*)stage;
    uint32_t* o32 = (uint32_t*)o;
    const uint64_t nwords = nbytes / 4;
    for (uint64_t k = threadIdx.x; k < nwords; k += BLOCK) o32[k] = s32[k];
    for (uint64_t k = nwords * 4 + threadIdx.x; k < nbytes; k += BLOCK) o[k] = stage[k];
    __syncthreads();
  }
}

// gleval.NormalsCentralDiff (gleval/gleval.go:53-108); h = step/2.
__global__ void __launch_bounds__(BLOCK) normals_kernel(const uint32_t* __restrict__ code_g, const float* __restrict__ pos,
                                                        float* __restrict__ nrm, uint64_t n, float h) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * BLOCK;
  for (uint64_t base = (uint64_t)blockIdx.x * BLOCK; base < n; base += step) {
    const uint64_t i = base + threadIdx.x;
    const bool valid = i < n;
    float px = 0, py = 0, pz = 0;
    if (valid) { px = pos[3 * i]; py = pos[3 * i + 1]; pz = pos[3 * i + 2]; }
    float out[3];
#pragma unroll 1
    for (int dim = 0; dim < 3; dim++) {
      P3 a = {px + (dim == 0 ? h : 0.f), py + (dim == 1 ? h : 0.f), pz + (dim == 2 ? h : 0.f)};
      P3 b = {px - (dim == 0 ? h : 0.f), py - (dim == 1 ? h : 0.f), pz - (dim == 2 ? h : 0.f)};
      P3 ab[2] = {a, b};
      float dd[2];
      gsdf_dev::sdf_eval<2>(code, ab, dd, lds, BLOCK);
      const float v = dd[0] - dd[1];
      if (dim == 0) out[0] = v; else if (dim == 1) out[1] = v; else out[2] = v;
    }
    if (valid) { nrm[3 * i] = out[0]; nrm[3 * i + 1] = out[1]; nrm[3 * i + 2] = out[2]; }
  }
}

