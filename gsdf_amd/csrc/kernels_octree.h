// kernels_octree.h -- the octree mesher: centre tests of the cube levels (prune_kernel; the speculative top: prune_spec_kernel +
// prune_resolve_kernel), the leaf phase (leaf_eval_kernel -> cut-leaf records -> march_records_kernel; the fused leaf_kernel and
// the corner-sharing leaf_brick_kernel) and the marching-cubes emission helpers the flat renderer's kernels use too.
//   glrender/octreerenderer.go:43-284, glrender/marchcubes.go:14-98
#pragma once
#include "kernels_common.h"

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
                                                      MeshCounters* __restrict__ ctr, const unsigned* __restrict__ spec_part,
                                                      unsigned spec_rows, int spec_top_S, unsigned spec_mask) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  // The first launch behind the speculative top also adds up that stage's statistics: prune_resolve_kernel leaves them as one
  // row of 16 counts per workgroup (items and passes of up to 8 levels) instead of issuing 14 atomics per workgroup on two
  // cache lines (1 000 of its 1 200 atomics, 9 of its 14 us). One workgroup, off the other workgroups' critical path.
  if (spec_part != nullptr && blockIdx.x == 0) {  // block-uniform
    __shared__ unsigned s_sum[16];
    if (threadIdx.x < 16u) s_sum[threadIdx.x] = 0u;
    __syncthreads();
    unsigned acc = 0;  // column threadIdx.x & 15 of rows threadIdx.x >> 4, + 16, ...: independent loads (a serial walk of the rows by 16 threads took 15 us)
    for (unsigned r = threadIdx.x >> 4; r < spec_rows; r += BLOCK / 16) acc += spec_part[r * 16u + (threadIdx.x & 15u)];
    if (acc) atomicAdd(&s_sum[threadIdx.x & 15u], acc);
    __syncthreads();
    if (threadIdx.x < 16u) {
      const int j = (int)(threadIdx.x >> 1), top = spec_top_S & 0xff, S = spec_top_S >> 8, lv = top - j;
      if (j < S) {
        if (threadIdx.x & 1u) ctr->n_pass[lv] = (unsigned long long)s_sum[threadIdx.x];
        else if (lv >= 3 && ((spec_mask >> lv) & 1u) != 0u) ctr->n_items[lv] = (unsigned long long)s_sum[threadIdx.x];
      }
    }
  }
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
        keep = valid && !nb::abs_ge(dv[0], maxDist);
      } else {
        // interval evaluation over the cube's bounding ball; `fired` = this cube's brick mask (dev_ops.h: D_SKIP), handed on in
        // Cube.w: the leaf kernels read it at the last level (there the cube IS the brick a wave evaluates), the levels above ignore it
        uint32_t fired = 0u;
        gsdf_dev::sdf_eval<2, 0, true>(code, pv, dv, lds, BLOCK, false, maxDist, (uint32_t)lip_base, 0u, &fired);
        keep = valid && !(nb::ge0(dv[0]) || nb::le0(dv[1]));
        c.w = (uint16_t)fired;
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
  // statistics. "Passed the test" differs from "kept" (n_level, counted by the flushes) only at the level where bricks are
  // dealt to ranks; everywhere else the host takes n_level for it and this atomic is not issued: a counter word serves ~70-90
  // returning or retiring atomics per microsecond, and one flush + one statistics atomic per workgroup were what the level
  // kernels' time consisted of (Level 3: 1024 workgroups, 2 x 1024 atomics, 24.6 us).
  if (shard_here) {
    if (lane == 0) s_w[4 + wave] = (unsigned)my_pass;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long t = (unsigned long long)s_w[4] + s_w[5] + s_w[6] + s_w[7];
      if (t) atomicAdd(&ctr->n_pass[level], t);
    }
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
                                                           uint8_t* __restrict__ pass, unsigned* __restrict__ clear_p, unsigned clear_words,
                                                           uint16_t* __restrict__ mask16 /* brick masks of the block's last level (by number within the level), or null */,
                                                           unsigned mask_off /* first cube of that level */) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  (void)ncols;
  const unsigned step = gridDim.x * BLOCK;
  // This is the chain's first kernel and touches neither the counters nor the group sums: it clears them for the kernels behind
  // it (a memset in front of the chain, or behind the previous one, is a launch of its own: ~5 us of every mesh).
  for (unsigned k = blockIdx.x * BLOCK + threadIdx.x; k < clear_words; k += step) clear_p[k] = 0u;
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
      keep = !nb::abs_ge(dv[0], maxDist);
    } else {
      uint32_t fired = 0u;
      gsdf_dev::sdf_eval<2, 0, true>(code, pv, dv, lds, BLOCK, false, maxDist, (uint32_t)lip_base, 0u, &fired);  // (maxDist differs from lane to lane: fine, it is the lane's own radius)
      keep = !(nb::ge0(dv[0]) || nb::le0(dv[1]));
      if (mask16 != nullptr && valid && tested && i >= mask_off) mask16[i - mask_off] = (uint16_t)fired;
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
                                                              MeshCounters* __restrict__ ctr, unsigned* __restrict__ part,
                                                              const uint16_t* __restrict__ mask16 /* brick masks of level top - (S - 1), by number within the level; or null */) {
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
        if (slot < SPEC_STAGE) {
          Cube c = spec_cube(j, k);
          if (mask16 != nullptr) c.w = mask16[k];
          s_q[slot] = c;
        }
      }
    }
  }
  __syncthreads();
  const unsigned n = s_n;
  if (threadIdx.x == 0) {
    s_base = n ? atomicAdd(&ctr->n_level[top - (S - 1)], (unsigned long long)n) : 0ull;
    if (part != nullptr) {  // statistics as a row of counts; the next launch adds the rows up (prune_kernel)
      for (int j = 0; j < 8; j++) { part[blockIdx.x * 16u + 2u * j] = s_items[j]; part[blockIdx.x * 16u + 2u * j + 1u] = s_pass[j]; }
    } else {
      for (int j = 0; j < S && j < 8; j++) {
        const int level = top - j;
        const bool tested = level >= 3 && ((test_mask >> level) & 1u) != 0u;
        if (s_items[j] && tested) atomicAdd(&ctr->n_items[level], (unsigned long long)s_items[j]);
        if (s_pass[j]) atomicAdd(&ctr->n_pass[level], (unsigned long long)s_pass[j]);
      }
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

// (LEAF_MIN_COLS and TRI_STAGE, which the host's LDS arithmetic needs too, are in kernels_common.h)

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
        index |= (nb::lt0(dk[kp]) ? 1u : 0u) << c;
      }
      if (c0 == 0) {
        pass = valid && nb::abs_le(dk[0], cubeDiag);
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
#define MARCH_GROUP 64              // blocks per entry of the group sums (records, triangles, active leaves) the evaluating kernel accumulates
// One 64-bit word per group: records in bits 0..19 (<= 64 blocks x 64 = 4096), triangles in bits 20..35 (<= 5 per record: 20 480),
// leaves that passed the corner-0 test in bits 36..49 (<= 4096), z rows evaluated in bits 50..63 (distinct-row bricks, DZ: <= 8
// per block, 512): the fields cannot carry into each other. The statistics ride on the
// same fire-and-forget atomic as the offsets -- spread over thousands of words -- because as three atomics per workgroup on
// two counter lines they were what bounded the kernel for a cheap tree: 16 384 workgroups x 3 at ~9 ns each = npt-flange's
// whole 0.46 ms (without them 0.42 ms, and the instruction savings of this round finally show).
#define PSUM_PACK(r, t, a, z) ((unsigned long long)(r) | ((unsigned long long)(t) << 20) | ((unsigned long long)(a) << 36) | ((unsigned long long)(z) << 50))
#define PSUM_REC(v) ((unsigned)((v) & 0xfffffull))
#define PSUM_TRI(v) ((unsigned)(((v) >> 20) & 0xffffull))
#define PSUM_ACT(v) ((unsigned)(((v) >> 36) & 0x3fffull))
#define PSUM_ROWS(v) ((unsigned)((v) >> 50))

// UCUBE: lq == 3 (every mesh of three levels or more): the 64 leaves of a wave pass are one level-3 cube.
// NTLDS: the triangles-per-case table sits in LDS behind the interpreter's columns; false when exactly those 256 bytes would
// cost a workgroup per CU (the host decides): the counts are then read from the table in global memory. A template argument
// and not a run-time flag: a select between an LDS and a global load makes the compiler form a flat pointer, and ROCm
// 7.0-7.2's backend then dies on some trees ("Illegal instruction detected ... V_CMP_NE_U32_e32 0, $src_shared_base").
// BOTH (specialised builds of programs with work that depends on x and y alone): the two passes of a column brick are laid out
// in one body, so the compiler's value numbering computes such subexpressions once per lane for all eight z instead of once per
// pass (npt-flange: hypot, atan2 and the thread's x,y terms; -4 % kernel time at the same register budget). Same statements on
// the same values.
// DZ (gsdf_mesh_opts.share_corners = 2; K = 4 column bricks): DISTINCT Z ROWS. The eight z rows of a brick are A0, A0+res, A1,
// A1+res, A2, ... with A_k = oz + res * (bz + k): row 2k-1 (the far corners of leaf layer k-1, (O + res*(i-1)) + res) and row 2k (the
// near corners of layer k, O + res*i) are the same plane, and the SAME FLOAT on 64-73 % of the planes at resdiv 1600. Equal bits
// in, equal bits out: a brick has 5..8 distinct rows (wave-uniform: the z of a brick is), only those are evaluated -- four in a
// first pass, the remaining 1, 2 or 4 in a second pass with as many points per lane -- and every leaf corner reads the row that
// holds its coordinate's value. Same distances, same records, same triangles; 24 % fewer evaluations on average. The reference
// evaluates every corner of every leaf (marchcubes.go:24-31), so this is an option, not the default; the rows evaluated travel in
// the group sums (statistics: MeshCounters.n_points).
template <int K, int WAVES, bool UCUBE = true, bool NTLDS = true, bool BOTH = false, bool DZ = false>
__global__ void __launch_bounds__(BLOCK, WAVES) leaf_eval_kernel(const uint32_t* __restrict__ code_g, const Cube* __restrict__ cubes,
                                                          unsigned long long cube_cap, int lq, int nslots, float ox, float oy, float oz,
                                                          float res, uint32_t* __restrict__ hdr, uint32_t* __restrict__ rec,
                                                          unsigned long long* __restrict__ psum, unsigned long long n_blocks_cap,
                                                          MeshCounters* __restrict__ ctr,
                                                          unsigned mask_valid /* the cubes' w fields are brick masks of the last centre-test level (dev_ops.h: D_SKIP) */) {
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
  unsigned my_cont = 0;  // wave-uniform (SGPR); only the leaf-per-lane form (!UCUBE) counts it: a column brick always goes on
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
    unsigned nact = 0;  // leaves of this wave pass that pass the corner-0 test (wave-uniform)
    unsigned zrows = 8u;  // z rows of the brick that were evaluated (wave-uniform; fewer than 8 with DZ)
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
      // what the last centre test proved for this whole brick: operand subtrees that cannot matter anywhere in it (scalar)
      const uint32_t bmask = mask_valid ? (__builtin_amdgcn_readfirstlane((uint32_t)(cw >> 48)) | GSDF_BRICK_MASK_VALID) : 0u;
      float dall[8];  // distances of rows 0..7; static shift register
#pragma unroll
      for (int j = 0; j < 8; j++) dall[j] = 0.f;
      gsdf_dev::XYCache xyc;  // what this lane's column knows from its other pass (one brick: reset per brick)
#define GSDF_COLUMN_PASS                                                     \
  {                                                                          \
    P3 pk[K];                                                                \
    float dk[K];                                                             \
    _Pragma("unroll") for (int kp = 0; kp < K; kp++) {                       \
      const unsigned r = c0 + kp; /* wave-uniform */                         \
      const float za = oz + res * (float)(uint16_t)(bz + (r >> 1));          \
      pk[kp].x = px;                                                         \
      pk[kp].y = py;                                                         \
      pk[kp].z = (r & 1u) ? za + res : za;                                   \
    }                                                                        \
    gsdf_dev::sdf_eval<K, 2>(code, pk, dk, lds, BLOCK, /*brick=*/true, 0.0f, 0u, bmask, nullptr, &xyc); \
    _Pragma("unroll") for (int j = 0; j < 8 - K; j++) dall[j] = dall[j + K]; \
    _Pragma("unroll") for (int kp = 0; kp < K; kp++) dall[8 - K + kp] = dk[kp]; \
  }
      unsigned rowmask = 0xffu, nrows = 8u;  // rows evaluated (bit r), wave-uniform
      if (DZ && K == 4) {
        // rows 2k-1 and 2k (k = 1..3): the same float? (bits, not ==: -0 and +0 are different points to an evaluator)
        unsigned mism = 0u;
#pragma unroll
        for (unsigned k = 1; k < 4; k++) {
          const float far = (oz + res * (float)(uint16_t)(bz + k - 1u)) + res, near = oz + res * (float)(uint16_t)(bz + k);
          mism |= (__float_as_uint(far) != __float_as_uint(near) ? 1u : 0u) << (k - 1u);
        }
        mism = __builtin_amdgcn_readfirstlane(mism);
#ifdef GSDF_EXP_DZ_FORCE  // developer experiment (GSDF_HIP_SPEC_FLAGS=-DGSDF_EXP_DZ_FORCE=0 / 7): every brick with 5 / 8 rows (timing only, wrong meshes)
        mism = GSDF_EXP_DZ_FORCE;
#endif
        rowmask = 0xabu | ((mism & 1u) << 2) | ((mism & 2u) << 3) | ((mism & 4u) << 4);  // rows 0, 1, 3, 5, 7 always; 2, 4, 6 where they differ
        nrows = (unsigned)__builtin_popcount(rowmask);
        zrows = nrows;
        // slot s (s-th distinct row) -> its row number, a nibble each; slots beyond nrows repeat row 7 (evaluated with a tail pass, never read)
        unsigned rowsw = 0u, mr = rowmask;
#pragma unroll
        for (unsigned sl = 0; sl < 8; sl++) {
          const unsigned r = mr ? (unsigned)__builtin_ctz(mr) : 7u;
          mr &= mr - 1u;
          rowsw |= r << (4u * sl);
        }
#define GSDF_ROWS_PASS(KK, NIB0, D0)                                                        \
  {                                                                                          \
    P3 pk[KK];                                                                               \
    float dk[KK];                                                                            \
    _Pragma("unroll") for (int kp = 0; kp < KK; kp++) {                                      \
      const unsigned r = (rowsw >> (4u * ((NIB0) + (unsigned)kp))) & 7u; /* wave-uniform */  \
      const float za = oz + res * (float)(uint16_t)(bz + (r >> 1));                          \
      pk[kp].x = px;                                                                         \
      pk[kp].y = py;                                                                         \
      pk[kp].z = (r & 1u) ? za + res : za;                                                   \
    }                                                                                        \
    gsdf_dev::sdf_eval<KK, 2>(code, pk, dk, lds, BLOCK, /*brick=*/true, 0.0f, 0u, bmask, nullptr, &xyc); \
    _Pragma("unroll") for (int kp = 0; kp < KK; kp++) dall[(D0) + kp] = dk[kp];              \
  }
        if (BOTH) {  // one body: what depends on x and y alone is computed once for all the rows
          GSDF_ROWS_PASS(4, 0u, 0)
          if (nrows <= 5u) GSDF_ROWS_PASS(1, 4u, 4)
          else if (nrows == 6u) GSDF_ROWS_PASS(2, 4u, 4)
          else GSDF_ROWS_PASS(4, 4u, 4)
        } else {  // one site per pass width (the interpreter's body is large): four rows once or twice, then a pass of two
          const unsigned np4 = nrows > 6u ? 2u : 1u;
#pragma unroll 1
          for (unsigned pz = 0; pz < np4; pz++) {
#pragma unroll
            for (int j = 0; j < 4; j++) dall[j] = dall[j + 4];
            GSDF_ROWS_PASS(4, 4u * pz, 4)
          }
          if (np4 == 1u) {
#pragma unroll
            for (int j = 0; j < 4; j++) dall[j] = dall[j + 4];
            GSDF_ROWS_PASS(2, 4u, 4)
          }
        }
#undef GSDF_ROWS_PASS
      } else if (BOTH) {
#ifdef GSDF_EXP_ROWS  // developer experiment (GSDF_HIP_SPEC_FLAGS=-DGSDF_EXP_ROWS=4): only the first rows of a brick -- what a row costs (timing only, wrong meshes)
#pragma unroll
        for (unsigned c0 = 0; c0 < GSDF_EXP_ROWS; c0 += K) GSDF_COLUMN_PASS
#else
#pragma unroll
        for (unsigned c0 = 0; c0 < 8; c0 += K) GSDF_COLUMN_PASS
#endif
      } else {
#pragma unroll 1
        for (unsigned c0 = 0; c0 < 8; c0 += K) GSDF_COLUMN_PASS
      }
#undef GSDF_COLUMN_PASS
      float* D = g_smem + (threadIdx.x & ~63u);  // rows of BLOCK floats; this wave's 64 columns of each
#pragma unroll
      for (int r = 0; r < 8; r++) D[r * BLOCK + lane] = dall[r];
      __builtin_amdgcn_wave_barrier();  // (the wave's LDS operations execute in order; this only pins the compiler's schedule)
      const unsigned li = lane & 3u, lj = (lane >> 2) & 3u, lk = lane >> 4;
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const unsigned cx = (c ^ (c >> 1)) & 1u, cy = (c >> 1) & 1u, cz = (c >> 2) & 1u;
        unsigned row = 2u * lk + cz;
        if (DZ && K == 4) row = (unsigned)__builtin_popcount(rowmask & ((2u << row) - 1u)) - 1u;  // the slot that holds this row's value (a row left out = the row before it)
        dc[c] = D[row * BLOCK + (2u * lj + cy) * 8u + 2u * li + cx];
        index |= (nb::lt0(dc[c]) ? 1u : 0u) << c;
      }
      pass = nb::abs_le(dc[0], cubeDiag);
      nact = (unsigned)__builtin_popcountll(__ballot(pass));
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
          index |= (nb::lt0(dk[kp]) ? 1u : 0u) << c;
        }
        if (c0 == 0) {
          pass = valid && nb::abs_le(dk[0], cubeDiag);
          const unsigned long long pmask = __ballot(pass);
          if (pmask == 0ull) break;  // wave-uniform
          const unsigned long long vmask = __ballot(valid);
          nact = (unsigned)__builtin_popcountll(pmask);
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
        if (DZ && UCUBE && K == 4) atomicAdd(&psum[blk / MARCH_GROUP], PSUM_PACK(nrec, ntri, nact, zrows));  // (every brick: its rows are counted)
        else if (nact) atomicAdd(&psum[blk / MARCH_GROUP], PSUM_PACK(nrec, ntri, nact, 0u));  // (a cut leaf is an active one: nrec <= nact)
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
  // statistics: active and cut leaves travel in the group sums (PSUM_PACK) and are totalled by march_records_kernel; "leaves
  // whose wave went on to the remaining corners" is every leaf for column bricks (the host knows) and counted here only for
  // the leaf-per-lane form of meshes with fewer than three levels (a handful of workgroups)
  if (!UCUBE) {
    unsigned* s_stat = (unsigned*)g_smem;
    __syncthreads();  // everyone is done with the interpreter columns
    if ((threadIdx.x & 63u) == 0u) s_stat[threadIdx.x >> 6] = my_cont;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long c = (unsigned long long)s_stat[0] + s_stat[1] + s_stat[2] + s_stat[3];
      if (c) atomicAdd(&ctr->n_cont, c);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// leaf_dense_kernel (gsdf_mesh_opts.share_corners = 1, two-kernel leaf phase): every BITWISE-DISTINCT lattice point of a brick once.
// A brick's 512 corner evaluations are the product of eight x, eight y and eight z coordinates of which, per axis, rows 2k-1 and
// 2k -- (O + res (i-1)) + res and O + res i -- are the same plane and mostly the same float (see DZ above): nx x ny x nz distinct
// points, 5..8 per axis, ~250 of 512 on average. The wave evaluates exactly those, packed four to a lane (unrelated points: no
// column sharing -- this pays where the field costs more per point than per (x, y) column: threads, knurls, transformed parts; a
// part that is mostly axisymmetric loses more through the sharing it gives up), the tail in a pass of
// one or two points per lane (TAILS; the interpreter build rounds up to passes of four), into 512 floats of LDS per wave; every
// leaf then reads its eight corners where its coordinates' values sit, and from there on the kernel is leaf_eval_kernel: the same
// records, headers and group sums for march_records_kernel / the packing kernels. Same distances on the same points, same bits.
// LDS: [max(nslots * 4, 8) columns of BLOCK floats | 256 B case counts if NTLDS | 4 x 512 distances | 4 x 24 coordinates | 4 words].
template <int WAVES, bool NTLDS, bool TAILS>
__global__ void __launch_bounds__(BLOCK, WAVES) leaf_dense_kernel(const uint32_t* __restrict__ code_g, const Cube* __restrict__ cubes,
                                                                  unsigned long long cube_cap, int nslots, float ox, float oy, float oz, float res,
                                                                  uint32_t* __restrict__ hdr, uint32_t* __restrict__ rec,
                                                                  unsigned long long* __restrict__ psum, unsigned long long n_blocks_cap,
                                                                  MeshCounters* __restrict__ ctr, unsigned mask_valid /* see leaf_eval_kernel */) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  const size_t cols = (size_t)(nslots * 4 > 8 ? nslots * 4 : 8) * BLOCK;
  uint8_t* s_nt = (uint8_t*)(g_smem + cols);
  float* s_D = g_smem + cols + (NTLDS ? 64 : 0);
  float* s_val = s_D + 4 * 512;
  unsigned* s_pts = (unsigned*)(s_val + 4 * 24);
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  float* D = s_D + wave * 512;
  float* val = s_val + wave * 24;
  if (NTLDS) s_nt[threadIdx.x] = GSDF_MC_NTRI[threadIdx.x];
  unsigned long long n_cubes = uniform_u64(ctr->n_level[3]);  // survivors of the last prune level (device-side count)
  if (NTLDS) __syncthreads();
  if (n_cubes > cube_cap) n_cubes = cube_cap;  // queue overflowed: host reruns with larger queues
  const float cubeDiag = 2 * 1.73205080757f * res;  // marchcubes.go:19
  const float org[3] = {ox, oy, oz};
  unsigned my_points = 0;  // wave-uniform
  const uint64_t step = (uint64_t)gridDim.x * 4;
  for (uint64_t brick = (uint64_t)blockIdx.x * 4 + wave; brick < n_cubes; brick += step) {  // wave-uniform; no barrier inside
    const unsigned long long cw = uniform_u64(*(const unsigned long long*)(cubes + brick));
    const unsigned pidx[3] = {(unsigned)(cw & 0xffffu), (unsigned)((cw >> 16) & 0xffffu), (unsigned)((cw >> 32) & 0xffffu)};
    const uint32_t bmask = mask_valid ? ((uint32_t)(cw >> 48) | GSDF_BRICK_MASK_VALID) : 0u;  // the brick's mask from the last centre test (scalar)
    // per axis: planes 1..3 with two distinct floats (bits, not ==: -0 and +0 are different points to an evaluator)
    unsigned mb[3], nax[3];
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
      const unsigned i0 = pidx[ax] * 4u;
      unsigned m = 0;
#pragma unroll
      for (unsigned k = 1; k < 4; k++) {
        const float far = (org[ax] + res * (float)(uint16_t)(i0 + k - 1u)) + res, near = org[ax] + res * (float)(uint16_t)(i0 + k);
        m |= (__float_as_uint(far) != __float_as_uint(near) ? 1u : 0u) << (k - 1u);
      }
      mb[ax] = __builtin_amdgcn_readfirstlane(m);
      nax[ax] = 5u + (unsigned)__builtin_popcount(mb[ax]);
    }
    if (lane < 12u) {  // coordinate table: lane (axis, a) writes A_a and A_a + res at their distinct-value slots (a merged plane is written twice, with equal bits)
      const unsigned ax = lane >> 2, a = lane & 3u;
      const unsigned mm = ax == 0 ? mb[0] : (ax == 1 ? mb[1] : mb[2]);
      const float oo = ax == 0 ? ox : (ax == 1 ? oy : oz);
      const unsigned pi = ax == 0 ? pidx[0] : (ax == 1 ? pidx[1] : pidx[2]);
      const float Aa = oo + res * (float)(uint16_t)(pi * 4u + a);
      const unsigned u = a + (unsigned)__builtin_popcount(mm & ((1u << a) - 1u));
      val[ax * 8u + u] = Aa;
      val[ax * 8u + u + 1u] = Aa + res;  // Box max = origin + size
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned nx = nax[0], nxy = nax[0] * nax[1], N = nxy * nax[2];
    const float inx = 1.0f / (float)nx, inxy = 1.0f / (float)nxy;
#define GSDF_DENSE_PASS(KK, T0)                                                                  \
  {                                                                                               \
    P3 pk[KK];                                                                                    \
    float dk[KK];                                                                                 \
    _Pragma("unroll") for (int kp = 0; kp < KK; kp++) {                                           \
      unsigned t = (T0) + (unsigned)kp * 64u + lane;                                              \
      if (t >= N) t = N - 1u; /* idle slots re-evaluate the last point (result discarded) */      \
      const unsigned uz = (unsigned)(((float)t + 0.5f) * inxy);                                   \
      const unsigned r = t - uz * nxy;                                                            \
      const unsigned uy = (unsigned)(((float)r + 0.5f) * inx);                                    \
      const unsigned ux = r - uy * nx;                                                            \
      pk[kp] = P3{val[ux], val[8u + uy], val[16u + uz]};                                          \
    }                                                                                             \
    gsdf_dev::sdf_eval<KK>(code, pk, dk, lds, BLOCK, /*brick=*/true, 0.0f, 0u, bmask);            \
    _Pragma("unroll") for (int kp = 0; kp < KK; kp++) {                                           \
      const unsigned t = (T0) + (unsigned)kp * 64u + lane;                                        \
      if (t < N) D[t] = dk[kp];                                                                   \
    }                                                                                             \
    my_points += (unsigned)(KK) * 64u;                                                            \
  }
    unsigned t0 = 0;
    if (TAILS) {
#pragma unroll 1
      for (; t0 + 192u < N; t0 += 256u) GSDF_DENSE_PASS(4, t0)  // (a remainder of more than three rows of lanes: four points per lane)
      if (t0 + 64u >= N) { if (t0 < N) GSDF_DENSE_PASS(1, t0) }
      else if (t0 + 128u >= N) GSDF_DENSE_PASS(2, t0)
      else if (t0 < N) GSDF_DENSE_PASS(4, t0)
    } else {
#pragma unroll 1
      for (; t0 < N; t0 += 256u) GSDF_DENSE_PASS(4, t0)
    }
#undef GSDF_DENSE_PASS
    __builtin_amdgcn_wave_barrier();
    // this lane's leaf (a, b, c) and its corner 0 in the distinct-point lattice
    const unsigned la = lane & 3u, lb = (lane >> 2) & 3u, lc = lane >> 4;
    const unsigned ux0 = la + (unsigned)__builtin_popcount(mb[0] & ((1u << la) - 1u));
    const unsigned uy0 = lb + (unsigned)__builtin_popcount(mb[1] & ((1u << lb) - 1u));
    const unsigned uz0 = lc + (unsigned)__builtin_popcount(mb[2] & ((1u << lc) - 1u));
    const unsigned tb = ux0 + nx * uy0 + nxy * uz0;
    float dc[8];
    unsigned index = 0;
#pragma unroll
    for (unsigned c = 0; c < 8; c++) {
      dc[c] = D[tb + ((c ^ (c >> 1)) & 1u) + nx * ((c >> 1) & 1u) + nxy * ((c >> 2) & 1u)];
      index |= (nb::lt0(dc[c]) ? 1u : 0u) << c;
    }
    __builtin_amdgcn_wave_barrier();  // (the next brick rewrites D and the table)
    const bool pass = nb::abs_le(dc[0], cubeDiag);
    const unsigned nact = (unsigned)__builtin_popcountll(__ballot(pass));
    const bool cut = pass && index != 0u && index != 255u;
    const unsigned long long cm = __ballot(cut);
    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
    const uint64_t blk = brick;  // block = brick = 64 consecutive leaves of the queue's order
    if (blk < n_blocks_cap) {
      unsigned ntri = 0;
      if (cm != 0ull) {  // wave-uniform
        unsigned nt = 0u;  // 0..5
        if (NTLDS) { if (cut) nt = (unsigned)s_nt[index]; }
        else { if (cut) nt = (unsigned)GSDF_MC_NTRI[index]; }
        ntri = (unsigned)__builtin_popcountll(__ballot((nt & 1u) != 0u)) + 2u * (unsigned)__builtin_popcountll(__ballot((nt & 2u) != 0u)) +
               4u * (unsigned)__builtin_popcountll(__ballot((nt & 4u) != 0u));
      }
      if (lane == 0u) {
        const uint32_t nrec = (uint32_t)__builtin_popcountll(cm);
        hdr[blk] = nrec | (ntri << 8);
        if (nact) atomicAdd(&psum[blk / MARCH_GROUP], PSUM_PACK(nrec, ntri, nact, 0u));
      }
      if (cut) {
        const uint32_t lx = (pidx[0] << 2) + la, ly = (pidx[1] << 2) + lb, lz = (pidx[2] << 2) + lc;
        uint2* w = (uint2*)(rec + blk * REC_BLOCK + rank * REC_WORDS);  // 40-byte records: five 8-byte stores
        w[0] = make_uint2(__float_as_uint(dc[0]), __float_as_uint(dc[1]));
        w[1] = make_uint2(__float_as_uint(dc[2]), __float_as_uint(dc[3]));
        w[2] = make_uint2(__float_as_uint(dc[4]), __float_as_uint(dc[5]));
        w[3] = make_uint2(__float_as_uint(dc[6]), __float_as_uint(dc[7]));
        w[4] = make_uint2((lx & 0xffffu) | ((ly & 0xffffu) << 16), (lz & 0xffffu) | (index << 16));
      }
    }
  }
  // statistics: the points evaluated (lane slots, idle ones included), one atomic per workgroup
  __syncthreads();
  if (lane == 0u) s_pts[wave] = my_points;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = (unsigned long long)s_pts[0] + s_pts[1] + s_pts[2] + s_pts[3];
    if (t) atomicAdd(&ctr->n_points, t);
  }
}

// Inclusive prefix sum over the 64 lanes of a wave by DPP row shifts and row broadcasts: six v_add_u32 with a DPP operand (the
// compiler folds the move into the integer add) -- __shfl_up goes through ds_bpermute, an LDS round trip per step.
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return v;
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

// ---- the marching kernels' vertex -----------------------------------------------------------------------------------------
// The two corners of a marching-cubes edge differ in ONE coordinate. mcInterpolate (marchcubes.go:76-98) forms every coordinate
// as a + t (b - a); on the two axes where a = b that is a + t * (+0) = a + (t * 0), and only the third needs the subtraction,
// the product and the choice between the snapped endpoints. Same float operations on the same values as mc_interp above --
// also for a NaN or infinite t, which reaches the unchanged coordinates through t * 0 -- at a third of the arithmetic.
// Edge word (the kernels' LDS table: one 16-bit entry per triangle corner): ca | cb << 3 | axis << 6 | pa << 8, pa bit 0 / 1 / 2 =
// corner a sits at the max of the leaf's box in x / y / z (Box{origin, origin + size}: max = min + res).
constexpr __host__ __device__ __forceinline__ uint32_t march_edge_word(unsigned e) {
  const unsigned ca = GSDF_MC_PAIR_A(e), cb = GSDF_MC_PAIR_B(e);
  const unsigned pa = ((ca ^ (ca >> 1)) & 1u) | (((ca >> 1) & 1u) << 1) | (((ca >> 2) & 1u) << 2);
  const unsigned pb = ((cb ^ (cb >> 1)) & 1u) | (((cb >> 1) & 1u) << 1) | (((cb >> 2) & 1u) << 2);
  const unsigned d = pa ^ pb;  // exactly one bit
  const unsigned axis = d == 1u ? 0u : (d == 2u ? 1u : 2u);
  return ca | (cb << 3) | (axis << 6) | (pa << 8);
}
// col: the record's 8 corner distances + leaf origin (11 floats)
__device__ __forceinline__ void march_vertex(uint32_t ed, const float* col, float res, float& rx, float& ry, float& rz) {
  const float x0 = col[8], y0 = col[9], z0 = col[10];
  const float v1 = col[ed & 7u], v2 = col[(ed >> 3) & 7u];
  const float x1 = x0 + res, y1 = y0 + res, z1 = z0 + res;  // Box max = origin + size
  const float px = (ed & 0x100u) ? x1 : x0, py = (ed & 0x200u) ? y1 : y0, pz = (ed & 0x400u) ? z1 : z0;  // corner a
  const unsigned axis = (ed >> 6) & 3u;
  const bool on_x = axis == 0u, on_y = axis == 1u;
  const float c0 = on_x ? x0 : (on_y ? y0 : z0), c1 = on_x ? x1 : (on_y ? y1 : z1);
  const bool a_max = ((ed >> (8u + axis)) & 1u) != 0u;
  const float ac = a_max ? c1 : c0, bc = a_max ? c0 : c1;
  const float eps = 1e-12f;
  const bool k1 = dm::absf(0.f - v1) < eps, k2 = dm::absf(0.f - v2) < eps;
  float t = 0.5f;
  if (!k1 || !k2) t = (0.f - v1) / (v2 - v1);
  float rv = ac + t * (bc - ac);
  if (k1 && !k2) rv = ac;
  if (k2 && !k1) rv = bc;
  const float z = (k1 != k2) ? 0.0f : t * 0.0f;  // a snapped endpoint's coordinates are the corner's own; else a + t * (a - a)
  const float fx = px + z, fy = py + z, fz = pz + z;
  rx = on_x ? rv : fx;
  ry = on_y ? rv : fy;
  rz = (axis == 2u) ? rv : fz;
}

// The triangle table as the marching kernels keep it in LDS: [256][16] 16-bit entries -- entries 0..14 the EDGE WORDS of the
// row's triangle corners (march_edge_word: what a vertex needs of its edge, without a second lookup), entry 15 the row's
// triangle count.
static __device__ __constant__ const uint16_t GSDF_MARCH_EDGE_WORD[16] = {
    (uint16_t)march_edge_word(0), (uint16_t)march_edge_word(1), (uint16_t)march_edge_word(2),  (uint16_t)march_edge_word(3),
    (uint16_t)march_edge_word(4), (uint16_t)march_edge_word(5), (uint16_t)march_edge_word(6),  (uint16_t)march_edge_word(7),
    (uint16_t)march_edge_word(8), (uint16_t)march_edge_word(9), (uint16_t)march_edge_word(10), (uint16_t)march_edge_word(11), 0, 0, 0, 0};
__device__ __forceinline__ void march_load_table(uint16_t* s_tri) {
  for (int k = threadIdx.x; k < 256 * 16; k += BLOCK) {
    const int e = GSDF_MC_TRI[k >> 4][k & 15];
    s_tri[k] = (k & 15) == 15 ? (uint16_t)GSDF_MC_NTRI[k >> 4] : (e >= 0 ? GSDF_MARCH_EDGE_WORD[e & 15] : (uint16_t)0);
  }
}

struct __attribute__((packed, aligned(4))) MarchV3 { float x, y, z; };

// One chunk of at most BLOCK cut-leaf records, one per lane (rw: the lane's record, `has`: it has one), marched into triangles
// number out, out + 1, ...: returns the chunk's triangle count (block-uniform). march_dense_kernel's chunk body (packed records,
// after a gather). `prefetch` runs once the lane's record is in LDS -- the caller's load of its NEXT record, in flight while this
// chunk is marched (a dependent global load is ~2 us).
// Ends with a barrier: the caller may rewrite the columns, the owner list and s_misc.
//   s_col [BLOCK][11]: 8 distances + origin (odd stride: one record's values, read together by neighbouring lanes, sit in 11 banks)
//   s_own [5 * BLOCK]: triangle -> table offset (index*16 + 3*number) | record << 12
//   s_tri: the table of edge words (march_load_table);  s_misc[0..3]: wave sums
template <typename Prefetch>
__device__ __forceinline__ unsigned march_chunk_emit(const uint32_t (&rw)[10], bool has, Prefetch prefetch, float ox, float oy, float oz,
                                                     float res, float* s_col, uint32_t* s_own, const uint16_t* s_tri, unsigned* s_misc,
                                                     float* __restrict__ tris, unsigned long long out) {
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  unsigned index = 0;
  if (has) {
#pragma unroll
    for (int c = 0; c < 8; c++) s_col[threadIdx.x * 11u + c] = __uint_as_float(rw[c]);
    const uint32_t xy = rw[8], zi = rw[9];
    index = zi >> 16;
    // the leaf origin exactly as the evaluating kernel formed it
    s_col[threadIdx.x * 11u + 8u] = ox + res * (float)(xy & 0xffffu);
    s_col[threadIdx.x * 11u + 9u] = oy + res * (float)(xy >> 16);
    s_col[threadIdx.x * 11u + 10u] = oz + res * (float)(zi & 0xffffu);
  }
  prefetch();
  // owner list: prefix sum of the records' triangle counts
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
  MarchV3* dst = (MarchV3*)(tris + out * 9);
  const unsigned n3 = total * 3u;
#pragma unroll 2
  for (unsigned k = threadIdx.x; k < n3; k += BLOCK) {
    const unsigned t = k / 3u, j = k - 3u * t;
    const uint32_t o = s_own[t];
    MarchV3 r;
    march_vertex(s_tri[(o & 4095u) + (2u - j)], s_col + (o >> 12) * 11u, res, r.x, r.y, r.z);  // reversed winding (marchcubes.go:64-68)
    dst[k] = r;
  }
  __syncthreads();  // the next chunk rewrites the columns, the owner list and s_misc
  return total;
}

// The same for ONE WAVE and at most 64 records, with no workgroup barrier: the wave's own columns, owner list and output range
// (march_records_kernel: the four waves of a workgroup work through their own shares at their own pace -- the barriers of the
// workgroup-wide form made every chunk wait for its slowest wave: half of the kernel's wave cycles were waits, round 4's PMC).
// A wave's LDS operations execute in order: a wave_barrier only pins the compiler's schedule.
//   wcol [64][11], wown [5 * 64]: as above, for the wave's records; owner entries carry the lane (record) << 12
template <typename Prefetch>
__device__ __forceinline__ unsigned march_chunk_emit_wave(const uint32_t (&rw)[10], bool has, Prefetch prefetch, float ox, float oy, float oz,
                                                          float res, float* wcol, uint32_t* wown, float* wstage, const uint16_t* s_tri,
                                                          float* __restrict__ tris, unsigned long long out) {
  const unsigned lane = threadIdx.x & 63u;
  unsigned index = 0;
  if (has) {
#pragma unroll
    for (int c = 0; c < 8; c++) wcol[lane * 11u + c] = __uint_as_float(rw[c]);
    const uint32_t xy = rw[8], zi = rw[9];
    index = zi >> 16;
    wcol[lane * 11u + 8u] = ox + res * (float)(xy & 0xffffu);  // the leaf origin exactly as the evaluating kernel formed it
    wcol[lane * 11u + 9u] = oy + res * (float)(xy >> 16);
    wcol[lane * 11u + 10u] = oz + res * (float)(zi & 0xffffu);
  }
  prefetch();
  const unsigned nt = index ? (unsigned)s_tri[index * 16 + 15] : 0u;
  const unsigned ti = wave_incl_scan_u32(nt);
  const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)ti, 63);
  const unsigned first = ti - nt;
  for (unsigned k = 0; k < nt; k++) wown[first + k] = (index * 16u + 3u * k) | (lane << 12);
  __builtin_amdgcn_wave_barrier();
#ifndef GSDF_MARCH_VERTEX_PER_LANE
  // ONE TRIANGLE PER LANE, its 36 bytes handed to the neighbours through LDS so that the wave's stores are 16 bytes per lane and
  // contiguous (round 6). Round 5 measured the two plain forms against each other -- a vertex per lane (768-byte store
  // instructions, twice the instructions per vertex: the record's origin, the owner word and the table row are fetched per
  // vertex) 0.120 ms, a triangle per lane with its three 12-byte stores at a 36-byte stride 0.141 ms -- and concluded that the
  // kernel pays for its output stream. It paid for the SHAPE of it: a HIP copy kernel moving the same bytes (136 MB in, 245 MB out)
  // with 16-byte nontemporal stores runs at 6.9 TB/s on this part, with plain 16-byte stores at 5.2-5.5 (tools/ubench/copy_rate.hip,
  // profiles/r6a_copy_rate.txt). So: the triangle's arithmetic once per lane, nine floats into the wave's staging rows (stride 9:
  // conflict-free), and the wave writes the round's 64 x 36 bytes as three 1024-byte nontemporal store instructions (dword-
  // aligned: a triangle is 36 bytes) -- the triangles are written once and never read again by this mesh.
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  typedef float f4a __attribute__((ext_vector_type(4)));
  for (unsigned t0 = 0; t0 < total; t0 += 64u) {  // wave-uniform
    const unsigned t = t0 + lane;
    if (t < total) {
      const uint32_t o = wown[t];
      const float* col = wcol + (o >> 12) * 11u;
      const uint16_t* row = s_tri + (o & 4095u);
      float* st = wstage + lane * 9u;
      march_vertex(row[2], col, res, st[0], st[1], st[2]);  // reversed winding (marchcubes.go:64-68)
      march_vertex(row[1], col, res, st[3], st[4], st[5]);
      march_vertex(row[0], col, res, st[6], st[7], st[8]);
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned nd = (total - t0 < 64u ? total - t0 : 64u) * 9u;  // dwords of this round (wave-uniform)
    float* gdst = tris + (out + t0) * 9ull;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const unsigned idx = 4u * lane + 256u * (unsigned)r;
      if (idx + 4u <= nd) {
        const f4u v = *(const f4a*)(wstage + idx);
#ifdef GSDF_EXP_MARCH_NO_STORE  // developer experiment (library built with -D...): the kernel without its output stream (timing only)
        if (v.x == 1.2345678e-30f)
#endif
        __builtin_nontemporal_store(v, (f4u*)(gdst + idx));
      } else {
        for (unsigned j = idx; j < nd; j++) __builtin_nontemporal_store(wstage[j], gdst + j);
      }
    }
    __builtin_amdgcn_wave_barrier();  // (the next round rewrites the staging rows)
  }
#else
  MarchV3* dst = (MarchV3*)(tris + out * 9);
  const unsigned n3 = total * 3u;
  // (the round-5 form: one output vertex per lane, a store is 768 contiguous bytes)
#pragma unroll 1
  for (unsigned k = lane; k < n3; k += 64u) {
    const unsigned t = k / 3u, j = k - 3u * t;
    const uint32_t o = wown[t];
    MarchV3 r;
    march_vertex(s_tri[(o & 4095u) + (2u - j)], wcol + (o >> 12) * 11u, res, r.x, r.y, r.z);  // reversed winding (marchcubes.go:64-68)
    dst[k] = r;
  }
#endif
  __builtin_amdgcn_wave_barrier();  // (the next chunk rewrites the columns and the owner list)
  return total;
}

// Marching cubes over the cut-leaf records. NO atomic, NO staging, NO workgroup barrier in the loop.
//  * Where things go: the evaluating kernel left, per group of MARCH_GROUP blocks, the number of records and of triangles
//    (psum); every workgroup sums those (a few KB from L2), takes an equal share of the RECORDS -- a contiguous range of
//    blocks, cut at block granularity -- splits it into four equal parts, one per WAVE, and knows from the same sums where
//    every part's first triangle goes. Triangles appear in block order, record order, table order (a pure function of the
//    survivor queue's order).
//  * Who computes what: a wave takes its blocks 64 at a time (a wave-level prefix of their record counts), their records 64 at a
//    time (one per lane, 8 distances + origin into the wave's LDS columns); a prefix sum of the records' triangle counts gives an
//    owner list (triangle -> record, table row); then ONE OUTPUT VERTEX PER LANE: lane k of a round computes vertex k % 3 of
//    triangle k / 3 -- one interpolation, on the axis its edge runs along (march_vertex) -- and stores its 12 bytes at
//    out*36 + 12 k: one store instruction of a wave is 768 contiguous bytes. The next chunk's record and the next pass's block
//    counts are in flight while a chunk is marched.
//  (History: round 1 appended LDS stages of 896 triangles through one counter word: 0.158 ms. Known offsets + equal shares,
//  workgroup-wide chunks of 256 records between barriers, three interpolations and two 64-bit shifts per vertex: 0.100-0.115 ms,
//  33.5 M wave-instructions and half of the wave cycles waiting -- rounds 2-4. This form: round 5.)
// LDS: [table 256 x 16 u16 | per wave: 11 record columns of 64 floats, owner list 5 x 64, block prefix 66 | scratch] = 25.9 KB: 6 workgroups per CU
#define MARCH_WAVE_WORDS (64 * 11 + 5 * 64 + 68 + 64 * 9)
#define MARCH_LDS_BYTES (256 * 16 * 2 + 4 * MARCH_WAVE_WORDS * 4 + 8 + 24 * 8)
__global__ void __launch_bounds__(BLOCK, 4) march_records_kernel(const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ rec,
                                                              const unsigned long long* __restrict__ psum,
                                                              unsigned long long n_blocks_cap, int lq, float ox, float oy, float oz,
                                                              float res, float* __restrict__ tris, uint64_t tri_cap,
                                                              MeshCounters* __restrict__ ctr, MeshCounters* __restrict__ host_ctr) {
  uint16_t* s_tri = (uint16_t*)g_smem;
  float* s_wave = (float*)(s_tri + 256 * 16);
  unsigned long long* s_u64 = (unsigned long long*)(((uintptr_t)(s_wave + 4 * MARCH_WAVE_WORDS) + 7) & ~(uintptr_t)7);  // [0..3] scan, [4..18] found (five cut points x 3)
  const unsigned long long n_cubes_l = ctr->n_level[lq];  // (needed only once the group sums are in: the two trips to memory overlap)
  march_load_table(s_tri);
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;

  // ---- this workgroup's share of the records, and where its triangles go. The sums run over the groups of the arena's CAPACITY:
  // the words behind the live blocks' groups were cleared with the counters and add nothing.
  const uint64_t n_grp = (n_blocks_cap + MARCH_GROUP - 1) / MARCH_GROUP;
  const uint64_t per = (n_grp + BLOCK - 1) / BLOCK;  // groups per thread (contiguous)
  uint64_t e0 = (uint64_t)threadIdx.x * per, e1 = e0 + per;
  if (e0 > n_grp) e0 = n_grp;
  if (e1 > n_grp) e1 = n_grp;
  unsigned long long lr = 0, lt = 0, la = 0;
#pragma unroll 8
  for (uint64_t e = e0; e < e1; e++) {
    const unsigned long long v = psum[e];
    lr += PSUM_REC(v);
    lt += PSUM_TRI(v);
    la += (unsigned long long)PSUM_ACT(v) | ((unsigned long long)PSUM_ROWS(v) << 32);  // (the z rows evaluated ride in the high half: both sums stay below 2^32)
  }
  unsigned long long R, T, A = 0, Z = 0;
  const unsigned long long br = block_scan_u64(lr, s_u64, &R) - lr, bt = block_scan_u64(lt, s_u64, &T) - lt;
  const unsigned long long n_cubes = uniform_u64(n_cubes_l);
  const uint64_t n_leaves = n_cubes << (3 * (lq - 1));
  uint64_t n_blocks = (n_leaves + 63) >> 6;
  if (n_blocks > n_blocks_cap) n_blocks = n_blocks_cap;  // (the cube queue overflowed: the host reruns)
  if (blockIdx.x == 0) {  // the statistics the evaluating kernel sent along: cut leaves = records, active leaves, z rows evaluated (DZ)
    (void)block_scan_u64(la, s_u64, &A);
    Z = A >> 32;
    A &= 0xffffffffull;
    if (threadIdx.x == 0) { ctr->n_cut = R; ctr->n_active = A; if (Z) ctr->n_points = Z * 64ull; }  // (leaf_dense_kernel counts its points itself)
  }
  // This is the mesh's last kernel: its first workgroup hands the counters to the host itself (pinned, device-mapped memory;
  // visible when the kernel has completed) -- the D2H copy that used to follow cost 4 us plus the gap in front of it.
  if (host_ctr != nullptr && blockIdx.x == 0) {
    const unsigned long long* src = (const unsigned long long*)ctr;
    unsigned long long* dst = (unsigned long long*)host_ctr;
    for (unsigned k = threadIdx.x; k < (unsigned)(sizeof(MeshCounters) / 8); k += BLOCK) {
      unsigned long long v = src[k];
      if (k == (unsigned)(__builtin_offsetof(MeshCounters, n_tris) / 8)) v = T;
      if (k == (unsigned)(__builtin_offsetof(MeshCounters, n_cut) / 8)) v = R;
      if (k == (unsigned)(__builtin_offsetof(MeshCounters, n_active) / 8)) v = A;
      if (k == (unsigned)(__builtin_offsetof(MeshCounters, n_points) / 8) && Z) v = Z * 64ull;
      if (k == (unsigned)(__builtin_offsetof(MeshCounters, overflow) / 8) && T > tri_cap) v = 1ull;
      dst[k] = v;
    }
  }
  if (R == 0ull) return;  // no surface here (n_tris stays 0)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctr->n_tris = T;
    if (T > tri_cap) ctr->overflow = 1ull;  // the host learns the exact size and reruns
  }
  if (T > tri_cap) return;
  const unsigned long long X0 = R * blockIdx.x / gridDim.x, X1 = R * (blockIdx.x + 1ull) / gridDim.x;  // records [X0, X1)
  if (X0 == X1) return;
  // five cut points: the workgroup's share in four parts, wave w takes records [Xc(w), Xc(w + 1)) -- like the shares themselves,
  // cut at block granularity by the same rule, so the parts tile the share and the shares tile the mesh
  auto cut = [&](unsigned w) -> unsigned long long { return X0 + (X1 - X0) * w / 4ull; };
  // the group in which the running record count reaches X (X > 0): found by the one thread whose groups straddle it. ONE walk over
  // the thread's groups serves all five cut points, eight group sums in flight at a time (round 6: the walk used to run once per
  // cut point with an early exit, i.e. one dependent load per group -- up to 5 x 24 trips to the cache in front of the first record)
  {
    unsigned long long Xc[5];
    bool mine = false;
#pragma unroll
    for (unsigned w = 0; w < 5u; w++) {
      Xc[w] = cut(w);
      mine = mine || (br < Xc[w] && Xc[w] <= br + lr);
    }
    if (mine) {
      unsigned long long acc = br, tacc = bt;
      for (uint64_t e = e0; e < e1; e += 8) {
        unsigned long long v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = (e + u < e1) ? psum[e + u] : 0ull;
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const unsigned long long nr_ = PSUM_REC(v[u]);
#pragma unroll
          for (unsigned w = 0; w < 5u; w++)
            if (acc < Xc[w] && Xc[w] <= acc + nr_) { s_u64[4 + 3 * w] = e + u; s_u64[5 + 3 * w] = acc; s_u64[6 + 3 * w] = tacc; }
          acc += nr_;
          tacc += PSUM_TRI(v[u]);
        }
      }
    }
  }
  __syncthreads();  // (the last one: from here on every wave is on its own)
  // first block whose exclusive record prefix is >= X, and the triangles before it -- for the wave's own begin (cut w) and end
  // (cut w + 1): the 64 header words of the two groups are loaded together (two dependent trips to memory would be ~2 us each)
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
  unsigned long long cutX[2], cutE[2], cutAcc[2], cutT[2];
  uint32_t cutH[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const unsigned w = wave_u + (unsigned)i;
    cutX[i] = cut(w);
    cutE[i] = uniform_u64(s_u64[4 + 3 * w]); cutAcc[i] = uniform_u64(s_u64[5 + 3 * w]); cutT[i] = uniform_u64(s_u64[6 + 3 * w]);
    const uint64_t b = cutE[i] * MARCH_GROUP + lane;
    cutH[i] = (cutX[i] != 0ull && b < n_blocks) ? hdr[b] : 0u;
  }
  unsigned long long cutB[2], cutTb[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    unsigned long long B = 0, tb = 0;
    if (cutX[i] != 0ull) {
      const unsigned nr = cutH[i] & 255u, ntr = cutH[i] >> 8;
      const unsigned ir = wave_incl_scan_u32(nr), it = wave_incl_scan_u32(ntr);
      const unsigned long long m = __ballot(cutAcc[i] + (ir - nr) >= cutX[i]);
      if (m != 0ull) {
        const int j = __builtin_ctzll(m);
        B = cutE[i] * MARCH_GROUP + (unsigned)j;
        tb = cutT[i] + (unsigned)__builtin_amdgcn_readlane((int)(it - ntr), j);
      } else {
        B = (cutE[i] + 1) * MARCH_GROUP;
        tb = cutT[i] + (unsigned)__builtin_amdgcn_readlane((int)it, 63);
      }
    }
    cutB[i] = uniform_u64(B); cutTb[i] = uniform_u64(tb);
  }
  const unsigned long long b_begin = cutB[0];
  unsigned long long b_end = cutB[1], out = cutTb[0];
  if (b_end > n_blocks) b_end = n_blocks;

  float* wcol = s_wave + wave_u * MARCH_WAVE_WORDS;
  uint32_t* wown = (uint32_t*)(wcol + 64 * 11);
  unsigned* wpre = (unsigned*)(wown + 5 * 64);  // [65] exclusive prefix of the pass's record counts, [64] = their sum
  float* wstage = (float*)(wpre + 68);          // [64][9]: a round's triangles on their way out (16-byte aligned)
  unsigned nr_next = (b_begin + lane < b_end) ? (hdr[b_begin + lane] & 255u) : 0u;
#ifdef GSDF_EXP_MARCH_STARTUP_ONLY  // developer experiment: what the kernel costs before its first record (timing only)
  if (nr_next != 0xffffffffu) return;
#endif
  for (uint64_t b0 = b_begin; b0 < b_end; b0 += 64) {  // wave-uniform
    const unsigned nr = nr_next;
    {  // the next pass's counts: a dependent global load (~2 us) in flight while this pass is marched
      const uint64_t bn = b0 + 64 + lane;
      nr_next = bn < b_end ? (hdr[bn] & 255u) : 0u;
    }
    const unsigned incl = wave_incl_scan_u32(nr);
    wpre[lane] = incl - nr;
    const unsigned Rp = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);  // records of this pass (wave-uniform)
    __builtin_amdgcn_wave_barrier();
    // One record per lane in chunks of 64. The next chunk's record is fetched into registers BEFORE the current chunk is marched.
    uint32_t rw[REC_WORDS];
    auto fetch = [&](unsigned q) {
#pragma unroll
      for (int c = 0; c < REC_WORDS; c++) rw[c] = 0u;
      if (q < Rp) {
        // the block holding record q: largest j with wpre[j] <= q (blocks without records share their successor's prefix)
        unsigned lo = 0, hi = 64;
#pragma unroll
        for (int it = 0; it < 6; it++) {
          const unsigned mid = (lo + hi) >> 1;
          if (wpre[mid] <= q) lo = mid; else hi = mid;
        }
        struct __attribute__((packed, aligned(8))) Rec { uint32_t w[REC_WORDS]; };  // 40 bytes, 8-byte aligned: wide loads
        const Rec v = *(const Rec*)(rec + (b0 + lo) * REC_BLOCK + (q - wpre[lo]) * REC_WORDS);
#pragma unroll
        for (int c = 0; c < REC_WORDS; c++) rw[c] = v.w[c];
      }
    };
    fetch(lane);
    for (unsigned q0 = 0; q0 < Rp; q0 += 64u) {  // wave-uniform
      const unsigned q = q0 + lane;
      out += march_chunk_emit_wave(rw, q < Rp, [&] { fetch(q + 64u); }, ox, oy, oz, res, wcol, wown, wstage, s_tri, tris, out);
    }
    __builtin_amdgcn_wave_barrier();  // (the next pass rewrites the prefix)
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Packed cut-leaf records: the mesh as a rank hands it to a gather (gsdf_mesh_opts.payload = GSDF_PAYLOAD_RECORDS).
//
// A triangle is 36 bytes; the cut leaf it came from is a 40-byte record that yields 2.0 triangles on the configs' surfaces --
// 20 bytes per triangle. Over xGMI (one ~60 GB/s link per peer) an all-gather of npt-flange's 6.8 M triangles at resdiv 1600 is
// 245 MB of wire per rank, several times what a rank takes to mesh its share; as records it is 136 MB, and marching cubes runs
// AFTER the gather, on every rank that receives, over everybody's records (march_dense_kernel: HBM-bound, ~0.1 ms for the whole
// mesh). Same records, same arithmetic, same triangles as march_records_kernel makes of them; only their order differs.
//
// Payload of n records (dense_payload_bytes(n)):   [n x 40-byte records, in block order][u32 per chunk of DENSE_CHUNK records:
// the chunk's triangle count], padded to 8 bytes -- so a receiver knows where every chunk's triangles go without a pass of its own.
//   scan_groups_kernel   (one workgroup) exclusive prefix of the groups' record counts; totals to the counters and to the host
//   pack_records_kernel  the sparse record slots of leaf_eval_kernel -> the payload; chunk triangle counts by wave-level atomics
//   march_dense_kernel   marching cubes over a buffer holding the payloads of several ranks side by side (or of one)
// ---------------------------------------------------------------------------------------------------------------------
#define DENSE_CHUNK 256
#define DENSE_MAX_PARTS 64
__host__ __device__ __forceinline__ unsigned long long dense_payload_bytes(unsigned long long n) {
  return n * 40ull + ((((n + DENSE_CHUNK - 1ull) / DENSE_CHUNK) * 4ull + 7ull) & ~7ull);
}
struct DensePart {
  unsigned long long off;     // where this rank's payload starts in the buffer (bytes, a multiple of 8)
  unsigned long long n_recs;  // its records
  unsigned long long tri0;    // triangles of the parts before it
};
struct DenseParts {
  int n;
  int pad;
  DensePart p[DENSE_MAX_PARTS];
};

// Exclusive prefix of the record counts of the groups (psum) -> grp_base; totals -> ctr (n_cut, n_tris, n_active) and, as the
// mesh's counters for the host, host_ctr (pinned, device-mapped: visible when the chain has completed); the payload's chunk
// counts are cleared for pack_records_kernel's atomics. ONE workgroup of 1024 threads: a few thousand to a few hundred thousand
// 8-byte words from L2.
__global__ void __launch_bounds__(1024) scan_groups_kernel(const unsigned long long* __restrict__ psum, unsigned long long n_blocks_cap, int lq,
                                                           MeshCounters* __restrict__ ctr, unsigned long long* __restrict__ grp_base,
                                                           uint8_t* __restrict__ payload, unsigned long long rec_cap,
                                                           MeshCounters* __restrict__ host_ctr) {
  __shared__ unsigned long long s_r[16], s_t[16], s_a[16];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  unsigned long long n_cubes = uniform_u64(ctr->n_level[lq]);
  const uint64_t n_leaves = n_cubes << (3 * (lq - 1));
  uint64_t n_blocks = (n_leaves + 63) >> 6;
  if (n_blocks > n_blocks_cap) n_blocks = n_blocks_cap;
  const uint64_t n_grp = (n_blocks + MARCH_GROUP - 1) / MARCH_GROUP;
  const uint64_t per = (n_grp + 1023) / 1024;
  uint64_t e0 = (uint64_t)tid * per, e1 = e0 + per;
  if (e0 > n_grp) e0 = n_grp;
  if (e1 > n_grp) e1 = n_grp;
  unsigned long long lr = 0, lt = 0, la = 0;
  for (uint64_t e = e0; e < e1; e++) {
    const unsigned long long v = psum[e];
    lr += PSUM_REC(v); lt += PSUM_TRI(v); la += (unsigned long long)PSUM_ACT(v) | ((unsigned long long)PSUM_ROWS(v) << 32);  // (rows evaluated in the high half)
  }
  unsigned long long ir = lr, it = lt, ia = la;  // wave inclusive scans (only the records' is needed per thread; the others as totals)
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned rl = __shfl_up((unsigned)ir, off, 64), rh = __shfl_up((unsigned)(ir >> 32), off, 64);
    const unsigned tl = __shfl_up((unsigned)it, off, 64), th = __shfl_up((unsigned)(it >> 32), off, 64);
    const unsigned al = __shfl_up((unsigned)ia, off, 64), ah = __shfl_up((unsigned)(ia >> 32), off, 64);
    if (lane >= (unsigned)off) { ir += ((unsigned long long)rh << 32) | rl; it += ((unsigned long long)th << 32) | tl; ia += ((unsigned long long)ah << 32) | al; }
  }
  if (lane == 63u) { s_r[wave] = ir; s_t[wave] = it; s_a[wave] = ia; }
  __syncthreads();
  unsigned long long R = 0, T = 0, A = 0, before = 0;
  for (unsigned w = 0; w < 16u; w++) {
    if (w < wave) before += s_r[w];
    R += s_r[w]; T += s_t[w]; A += s_a[w];
  }
  const unsigned long long Z = A >> 32;  // z rows evaluated (DZ)
  A &= 0xffffffffull;
  unsigned long long acc = before + ir - lr;
  for (uint64_t e = e0; e < e1; e++) {
    grp_base[e] = acc;
    acc += PSUM_REC(psum[e]);
  }
  const bool fits = R <= rec_cap;
  if (fits) {
    uint32_t* chunk_tri = (uint32_t*)(payload + R * 40ull);
    const unsigned long long nch = (R + DENSE_CHUNK - 1ull) / DENSE_CHUNK;
    for (unsigned long long k = tid; k < ((nch + 1ull) & ~1ull); k += 1024) chunk_tri[k] = 0u;  // (+ the padding word)
  }
  if (tid == 0) {
    ctr->n_cut = R; ctr->n_tris = T; ctr->n_active = A; if (Z) ctr->n_points = Z * 64ull;
    if (!fits) ctr->overflow = 1ull;  // the host learns the exact size and reruns
  }
  if (host_ctr != nullptr) {
    const unsigned long long* src = (const unsigned long long*)ctr;
    unsigned long long* dst = (unsigned long long*)host_ctr;
    for (unsigned k = tid; k < (unsigned)(sizeof(MeshCounters) / 8); k += 1024) {
      unsigned long long v = src[k];
      if (k == (unsigned)(__builtin_offsetof(MeshCounters, n_tris) / 8)) v = T;
      if (k == (unsigned)(__builtin_offsetof(MeshCounters, n_cut) / 8)) v = R;
      if (k == (unsigned)(__builtin_offsetof(MeshCounters, n_active) / 8)) v = A;
      if (k == (unsigned)(__builtin_offsetof(MeshCounters, n_points) / 8) && Z) v = Z * 64ull;
      if (k == (unsigned)(__builtin_offsetof(MeshCounters, overflow) / 8) && !fits) v = 1ull;
      dst[k] = v;
    }
  }
}

// The record slots of leaf_eval_kernel (a 2 560-byte slot per 64-leaf block, its first hdr & 255 records used) -> the payload.
// One workgroup per group of MARCH_GROUP blocks (grid-stride), a wave per 16 of them: 8-byte copies, a lane per piece; then a
// lane per record adds its triangle count to its chunk's word -- per block at most two chunks, so two wave-level atomics.
__global__ void __launch_bounds__(BLOCK) pack_records_kernel(const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ rec,
                                                             const unsigned long long* __restrict__ grp_base, unsigned long long n_blocks_cap,
                                                             int lq, const MeshCounters* __restrict__ ctr, uint8_t* __restrict__ payload,
                                                             unsigned long long rec_cap) {
  __shared__ uint8_t s_nt[256];
  s_nt[threadIdx.x] = GSDF_MC_NTRI[threadIdx.x];
  __syncthreads();
  const unsigned long long R = uniform_u64(ctr->n_cut);
  if (R > rec_cap || R == 0ull) return;
  unsigned long long n_cubes = uniform_u64(ctr->n_level[lq]);
  const uint64_t n_leaves = n_cubes << (3 * (lq - 1));
  uint64_t n_blocks = (n_leaves + 63) >> 6;
  if (n_blocks > n_blocks_cap) n_blocks = n_blocks_cap;
  const uint64_t n_grp = (n_blocks + MARCH_GROUP - 1) / MARCH_GROUP;
  uint32_t* chunk_tri = (uint32_t*)(payload + R * 40ull);
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (uint64_t g = blockIdx.x; g < n_grp; g += gridDim.x) {
    const uint64_t b_lane = g * MARCH_GROUP + lane;
    const unsigned nr = b_lane < n_blocks ? (hdr[b_lane] & 255u) : 0u;
    unsigned incl = nr;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned u = __shfl_up(incl, off, 64);
      if (lane >= (unsigned)off) incl += u;
    }
    const unsigned long long base = uniform_u64(grp_base[g]);
    for (unsigned j = 0; j < 16u; j++) {
      const unsigned b = wave * 16u + j;
      const unsigned nrb = (unsigned)__shfl((int)nr, (int)b, 64);
      if (nrb == 0u) continue;  // wave-uniform
      const unsigned long long d0 = base + (unsigned long long)((unsigned)__shfl((int)incl, (int)b, 64) - nrb);  // first dense record of the block
      const uint32_t* src_w = rec + (g * MARCH_GROUP + b) * REC_BLOCK;
      const uint2* src = (const uint2*)src_w;
      uint2* dst = (uint2*)(payload + d0 * 40ull);
      for (unsigned i = lane; i < nrb * 5u; i += 64u) dst[i] = src[i];
      unsigned nt = 0u;
      bool second = false;
      const unsigned long long ch0 = d0 / DENSE_CHUNK;
      if (lane < nrb) {
        nt = (unsigned)s_nt[src_w[lane * REC_WORDS + 9u] >> 16];
        second = (d0 + lane) / DENSE_CHUNK != ch0;
      }
      const unsigned long long m1 = __ballot((nt & 1u) != 0u), m2 = __ballot((nt & 2u) != 0u), m4 = __ballot((nt & 4u) != 0u);
      const unsigned long long sm = __ballot(second);
      const unsigned sA = (unsigned)__builtin_popcountll(m1 & ~sm) + 2u * (unsigned)__builtin_popcountll(m2 & ~sm) + 4u * (unsigned)__builtin_popcountll(m4 & ~sm);
      const unsigned sB = (unsigned)__builtin_popcountll(m1 & sm) + 2u * (unsigned)__builtin_popcountll(m2 & sm) + 4u * (unsigned)__builtin_popcountll(m4 & sm);
      if (lane == 0u) {
        if (sA) atomicAdd(&chunk_tri[ch0], sA);
        if (sB) atomicAdd(&chunk_tri[ch0 + 1ull], sB);
      }
    }
  }
}

// Marching cubes over packed records: `buf` holds the payloads of parts->n ranks (a gather's receive buffer; or one rank's own).
// The chunks of all parts form one list; a workgroup takes an equal, contiguous share of it and knows from the chunk counts
// where its first triangle goes (the part's tri0 + the counts of the part's chunks before it: a few KB from L2). Triangles
// come out part-major, in record order. The chunk body is march_records_kernel's (march_chunk_emit): one record per lane, the
// next chunk's record in flight while this one is marched, one output vertex per lane.
// LDS: [11 record columns | owner list | table of edge words 8 KB | misc] = 24.7 KB: 6 workgroups per CU.
#define MARCH_DENSE_LDS_BYTES (11 * BLOCK * 4 + 5 * BLOCK * 4 + 256 * 16 * 2 + 8 * 4 + 8 + 4 * 8)
__global__ void __launch_bounds__(BLOCK, 6) march_dense_kernel(const uint8_t* __restrict__ buf, const DenseParts* __restrict__ parts, float ox, float oy, float oz,
                                                               float res, float* __restrict__ tris) {
  float* s_col = g_smem;
  uint32_t* s_own = (uint32_t*)(s_col + 11 * BLOCK);
  uint16_t* s_tri = (uint16_t*)(s_own + 5 * BLOCK);
  unsigned* s_misc = (unsigned*)(s_tri + 256 * 16);
  unsigned long long* s_u64 = (unsigned long long*)(((uintptr_t)(s_misc + 8) + 7) & ~(uintptr_t)7);
  march_load_table(s_tri);
  const int np = parts->n;
  auto chunks_of = [&](int p) -> unsigned long long { return (parts->p[p].n_recs + DENSE_CHUNK - 1ull) / DENSE_CHUNK; };
  unsigned long long G = 0;
  for (int p = 0; p < np; p++) G += chunks_of(p);
  const unsigned long long g0 = G * blockIdx.x / gridDim.x, g1 = G * (blockIdx.x + 1ull) / gridDim.x;
  if (g0 == g1) return;
  // the part and the chunk within it where this workgroup starts
  int p = 0;
  unsigned long long c = g0;
  while (p < np && c >= chunks_of(p)) { c -= chunks_of(p); p++; }
  // triangles before it
  unsigned long long before = 0;
  {
    const uint32_t* ct = (const uint32_t*)(buf + parts->p[p].off + parts->p[p].n_recs * 40ull);
    unsigned long long acc = 0;
    for (unsigned long long k = threadIdx.x; k < c; k += BLOCK) acc += ct[k];
    unsigned long long tot;
    (void)block_scan_u64(acc, s_u64, &tot);
    before = parts->p[p].tri0 + tot;
  }
  unsigned long long out = before;
  struct __attribute__((packed, aligned(8))) Rec { uint32_t w[REC_WORDS]; };
  uint32_t rw[REC_WORDS];
  bool has = false;
  auto fetch = [&](int fp, unsigned long long fc, bool in_range) {
#pragma unroll
    for (int k = 0; k < REC_WORDS; k++) rw[k] = 0u;
    has = false;
    if (in_range) {
      const unsigned long long q = fc * DENSE_CHUNK + threadIdx.x;
      if (q < parts->p[fp].n_recs) {
        const Rec v = *(const Rec*)(buf + parts->p[fp].off + q * 40ull);
#pragma unroll
        for (int k = 0; k < REC_WORDS; k++) rw[k] = v.w[k];
        has = true;
      }
    }
  };
  fetch(p, c, true);
  for (unsigned long long g = g0; g < g1; g++) {  // block-uniform
    // the chunk after this one (possibly the first of the next part that has any)
    int pn = p;
    unsigned long long cn = c + 1;
    while (pn < np && cn >= chunks_of(pn)) { cn = 0; pn++; }
    const bool more = g + 1 < g1;
    const bool has_now = has;
    const unsigned n = march_chunk_emit(rw, has_now, [&] { fetch(pn, cn, more); }, ox, oy, oz, res, s_col, s_own, s_tri, s_misc, tris, out);
    out += n;
    if (pn != p && pn < np) out = parts->p[pn].tri0;  // (the same number, unless a payload's counts are damaged)
    p = pn; c = cn;
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
    for (unsigned cc = 0; cc < 8; cc++) index |= (nb::lt0(vdist(cc)) ? 1u : 0u) << cc;
    const bool pass = bvalid && nb::abs_le(vdist(0), cubeDiag);
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
