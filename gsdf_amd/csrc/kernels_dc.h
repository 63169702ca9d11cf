// kernels_dc.h -- dual contouring on device (glrender/dual_contour.go:26-293, dual_contour_vertexplacement.go:26-223).
#pragma once
#include "kernels_common.h"

// =================================================================================================
// Dual contouring on device (glrender/dual_contour.go, dual_contour_vertexplacement.go).
// The reference keeps a map[i3.Vec]int over a full BFS decomposition; here the lattice is a dense
// int32 index grid in HBM (levels <= 11 -> <= 4.3 GB, trivial against 288 GB), so neighbour lookups
// are single loads and every stage is one lane per cube / per active edge.
// =================================================================================================
// The lists (kept cubes, active edges) are appended to with returning atomics, one per workgroup pass -- and ONE counter word
// serves only so many of them: ~88 per microsecond with the device to itself, 15-30 ns each when thousands of passes queue on it
// (round 5, npt-flange at resdiv 800: a second, idle atomic per pass on the same word took the origin sweep from 0.43 to 0.75 ms
// and the edge stage from 0.42 to 0.58 -- 10.8 K passes each; the text plate's 3.1 K did not notice). So each list is kept in
// DC_PARTS PARTS, every part with a counter word on a cache line of its own and its own eighth of the arrays; a workgroup appends
// to part (blockIdx.x % DC_PARTS), readers walk the parts one after the other (dc_flat_to_part).
#define DC_PARTS 8
#define DC_WORD_STRIDE 16  // unsigned long longs between two parts' counter words (128 bytes)
struct DCCounters {
  // kept cubes / active edges AND the runs they were appended in (DC_DESC below), one word per part: entries in the low 36 bits,
  // runs above them -- a pass reserves its entries and its run descriptor(s) with one atomic
  unsigned long long cubes_w[DC_PARTS * DC_WORD_STRIDE], edges_w[DC_PARTS * DC_WORD_STRIDE];
  unsigned long long n_tris, q_overflow, t_overflow;
  unsigned long long n_origin_evals;  // lattice cells whose origin was actually evaluated (the rest were outside the exact box) -- the host's total of:
  // the same count in 64 parts, one cache line each: every workgroup of the origin sweep adds its count once, and 8 192 adds to ONE
  // word were 0.3 of the sweep's 0.43 ms (round 5, rocprofv3: dc_origin_kernel 425 us for 8.7 M evaluations)
  unsigned long long n_origin_part[64 * 8];
};
#define DC_W_COUNT(w) ((w) & 0xfffffffffull)
#define DC_W_RUNS(w) ((w) >> 36)

// A RUN = cubes (edges) that one workgroup pass appended: contiguous in the list and neighbours in space (one 8 x 8 x 4K tile of the
// origin sweep; the edges of the passes over such a tile). The lists themselves are in the order the passes' atomics arrived -- runs
// from all over the part interleave -- and a wave that takes 64 consecutive entries straddles two or three runs from different
// places: every far-child gate of a wide union (dev_ops.h: D_GATE*) then stays shut for it and the polygon edge culling keeps most
// edges. (Round 5, text plate at resdiv 800: the normals stage issued ~2 500 lane instructions per evaluation, four glyph outlines
// in full; the edge stage 570.) So the evaluating stages take their work run by run: descriptor = first entry << 13 | entries
// (<= 1024 cubes of a tile; edge runs are cut into wave-sized pieces of <= 64). Every part has its own run arrays.
#define DC_DESC(first, count) (((unsigned long long)(first) << 13) | (unsigned long long)(count))
#define DC_DESC_FIRST(d) ((d) >> 13)
#define DC_DESC_COUNT(d) ((unsigned)((d) & 8191ull))

// Sizes of a part's share of the arrays, from the kernels' capacity arguments (the host sizes the arenas with the same functions).
//   cubes, distances, vertices, normals : cube_cap / DC_PARTS entries (cube_cap is a multiple of DC_PARTS)
//   cube runs                            : one per tile the part's workgroups sweep: tile T goes to workgroup T % gridDim and the grid
//                                          is a multiple of DC_PARTS, so to part T % DC_PARTS
//   edges                                : 3 per cube; edge runs: one per 64 edges and one more per cube run (its last, ragged one)
__host__ __device__ __forceinline__ unsigned long long dc_cube_run_seg(unsigned long long ntiles) { return (ntiles + DC_PARTS - 1) / DC_PARTS + 1; }
// The origin sweep's distances of the kept cubes (one float each) wait for the edge stage behind the cubes' run descriptors.
__host__ __device__ __forceinline__ unsigned long long dc_origin_dist_offset_words(unsigned long long ntiles) { return DC_PARTS * dc_cube_run_seg(ntiles); }
__host__ __device__ __forceinline__ unsigned long long dc_edge_run_seg(unsigned long long cube_cap, unsigned long long ntiles) {
  return 3 * (cube_cap / DC_PARTS) / 64 + dc_cube_run_seg(ntiles) + 8;
}
// Entry g of a list read as one (parts one after the other) -> its part and its place there; false past the end. n[] = entries per part.
__device__ __forceinline__ bool dc_flat_to_part(unsigned long long g, const unsigned long long (&n)[DC_PARTS], unsigned& part, unsigned long long& k) {
  unsigned p = 0;
  unsigned long long np = n[0];
#pragma unroll
  for (int q = 0; q < DC_PARTS - 1; q++) {
    if (p == (unsigned)q && g >= n[q]) { g -= n[q]; p = (unsigned)q + 1u; np = n[q + 1]; }
  }
  part = p; k = g;
  return g < np;
}
// the parts' counts (entries or runs) as block-uniform values, clipped to a part's capacity; their sum
template <bool RUNS>
__device__ __forceinline__ unsigned long long dc_part_counts(const unsigned long long* __restrict__ words, unsigned long long cap, unsigned long long (&n)[DC_PARTS]) {
  unsigned long long total = 0;
#pragma unroll
  for (int q = 0; q < DC_PARTS; q++) {
    const unsigned long long w = uniform_u64(words[q * DC_WORD_STRIDE]);
    unsigned long long v = RUNS ? DC_W_RUNS(w) : DC_W_COUNT(w);
    if (v > cap) v = cap;
    n[q] = v;
    total += v;
  }
  return total;
}

// The index grid's cells that are READ but not written by the origin sweep <- -1 ("no cube here"). The sweep writes every cell of
// the tiles it evaluates and steps over the tiles the block test cleared; the later stages look at the cells around kept cubes only
// (one cell either way: dc_place_kernel, dc_quads_kernel), i.e. at evaluated tiles and their 26 neighbours. So a cleared tile is
// filled iff one of its neighbours is evaluated -- the rest of the range is never read. (Until round 5 the whole range was filled:
// 213 MB and 39 us for npt-flange at resdiv 800.) flags: dc_block_test_kernel's, four per tile.
// A workgroup takes a block of 4 x 4 x 4 tiles, a tile per thread of its first wave: which tiles of the block and of the shell around it
// are evaluated goes through LDS (one load per thread; 27 dependent trips to memory per tile took 30 us whatever the lattice's size),
// the tiles to fill are listed, and the waves fill them side by side, a row of 8 cells (32 bytes) per lane and store pair.
__global__ void __launch_bounds__(BLOCK) dc_grid_clear_kernel(int* __restrict__ grid, int nshift, int K, unsigned zlo, unsigned zhi, unsigned tx0, unsigned ty0,
                                                              unsigned tz0, unsigned ntx, unsigned nty, unsigned ntz, const uint32_t* __restrict__ flags) {
  __shared__ unsigned char s_ev[6 * 6 * 6];
  __shared__ unsigned s_n;
  __shared__ unsigned s_list[64];
  const unsigned n = 1u << nshift;
  const unsigned nbx = (ntx + 3u) >> 2, nby = (nty + 3u) >> 2, nbz = (ntz + 3u) >> 2;
  const unsigned nblocks = nbx * nby * nbz;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (unsigned B = blockIdx.x; B < nblocks; B += gridDim.x) {  // block-uniform trip count
    const int bx = (int)(B % nbx) * 4, by = (int)((B / nbx) % nby) * 4, bz = (int)(B / (nbx * nby)) * 4;
    if (threadIdx.x < 216u) {
      const unsigned e = threadIdx.x;
      const int ux = bx + (int)(e % 6u) - 1, uy = by + (int)((e / 6u) % 6u) - 1, uz = bz + (int)(e / 36u) - 1;
      unsigned ev = 0u;
      if (ux >= 0 && uy >= 0 && uz >= 0 && ux < (int)ntx && uy < (int)nty && uz < (int)ntz) {
        const uint4 f = *(const uint4*)(flags + ((uint64_t)ux + (uint64_t)ntx * ((uint64_t)uy + (uint64_t)nty * (uint64_t)uz)) * 4ull);
        ev = (f.x | f.y | f.z | f.w) >> 31;
      }
      s_ev[e] = (unsigned char)ev;
    }
    if (threadIdx.x == 0) s_n = 0u;
    __syncthreads();
    if (threadIdx.x < 64u) {
      const unsigned lx = threadIdx.x & 3u, ly = (threadIdx.x >> 2) & 3u, lz = threadIdx.x >> 4;
      const unsigned x = (unsigned)bx + lx, y = (unsigned)by + ly, z = (unsigned)bz + lz;
      if (x < ntx && y < nty && z < ntz && s_ev[(lz + 1u) * 36u + (ly + 1u) * 6u + lx + 1u] == 0u) {  // (an evaluated tile is written by the sweep)
        unsigned near = 0u;
#pragma unroll
        for (unsigned dz = 0; dz < 3u; dz++)
#pragma unroll
          for (unsigned dy = 0; dy < 3u; dy++)
#pragma unroll
            for (unsigned dx = 0; dx < 3u; dx++) near |= s_ev[(lz + dz) * 36u + (ly + dy) * 6u + lx + dx];
        if (near != 0u) s_list[atomicAdd(&s_n, 1u)] = x | (y << 10) | (z << 20);  // (tile coordinates are below 2^10: 2^11 cells / 8, / 4)
      }
    }
    __syncthreads();
    // a wave per tile found: 8 x 4K rows of 8 cells, a row per lane and step (the rows start on 32-byte boundaries: two 16-byte stores)
    const unsigned cnt = s_n;
    for (unsigned q = wave; q < cnt; q += 4u) {
      const unsigned F = s_list[q];
      const unsigned x = F & 1023u, y = (F >> 10) & 1023u, z = F >> 20;
      const unsigned cx = (tx0 + x) * 8u;
      if (cx >= n) continue;  // (n is a multiple of 8: a row lies in the lattice or it does not)
      for (unsigned r = lane; r < 32u * (unsigned)K; r += 64u) {
        const unsigned cy = (ty0 + y) * 8u + (r & 7u), cz = zlo + (tz0 + z) * (4u * (unsigned)K) + (r >> 3);
        if (cy < n && cz < zhi) {
          int4* row = (int4*)(grid + ((uint64_t)cx + ((uint64_t)cy << nshift) + ((uint64_t)cz << (2 * nshift))));
          row[0] = make_int4(-1, -1, -1, -1);
          row[1] = make_int4(-1, -1, -1, -1);
        }
      }
    }
    __syncthreads();  // (s_ev and s_list are rewritten by the next pass)
  }
}

// Stage 0 (round 5): which blocks of the origin sweep can hold a kept cube at all. The sweep's work item is a wave's block of
// 8 x 8 x K cell origins; a cube is kept iff |d(origin)| < 2 res, and the reference evaluates every origin of its cubic lattice
// to find that out (dual_contour.go:26-83). Here one lane per block evaluates the field in INTERVAL mode (interp.h: LIP, what the
// octree's centre tests run) over the ball that holds the block's origins: bounds that exclude (-2 res, 2 res) by the margin prove
// "nothing kept here" for all 64 K origins at the price of two evaluations, and the sweep skips the block -- same kept cubes, same
// grid, a fraction of the evaluations (the kept cubes hug the surface; the sweep covers the part's whole box).
// The same evaluation proves more than that: which operand subtrees of the program cannot matter anywhere in the block -- its BRICK
// MASK (dev_ops.h: D_SKIP), which the sweep hands its evaluator as the octree's last centre test does for the leaf kernels.
// flags[T * 4 + w]: bit 31 = block w of tile T must be evaluated, bits 0..15 = its brick mask. LDS: [2 * ncols floats per lane] as prune_kernel.
__global__ void __launch_bounds__(BLOCK) dc_block_test_kernel(const uint32_t* __restrict__ code_g, int ncols, int lip_base, int K, float ox, float oy, float oz,
                                                              float res, unsigned zlo, unsigned tx0, unsigned ty0, unsigned tz0, unsigned ntx, unsigned nty,
                                                              unsigned ntz, uint32_t* __restrict__ flags) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  (void)ncols;
  const uint64_t nblk = (uint64_t)ntx * nty * ntz * 4ull;
  const float kf = (float)(K - 1);
  // the block's origins span 7 x 7 x (K - 1) cells: half its diagonal, a little more for the rounding of the coordinates
  const float radius = 0.5f * res * dm::sqrtf_(98.0f + kf * kf) * 1.0001f;
  const float keepDist = res * 2 * 1.001f;
  for (uint64_t b0 = (uint64_t)blockIdx.x * BLOCK; b0 < nblk; b0 += (uint64_t)gridDim.x * BLOCK) {  // block-uniform trip count
    const uint64_t b = b0 + threadIdx.x;
    const bool valid = b < nblk;
    const unsigned T = (unsigned)((valid ? b : 0ull) >> 2);  // (tile counts stay below 2^32: 32-bit divisions)
    const unsigned w = (unsigned)(b & 3ull);
    const unsigned tx = tx0 + T % ntx, ty = ty0 + (T / ntx) % nty, tz = tz0 + T / (ntx * nty);
    P3 c;
    c.x = ox + res * ((float)(tx * 8u) + 3.5f);
    c.y = oy + res * ((float)(ty * 8u) + 3.5f);
    c.z = oz + res * ((float)(zlo + tz * (4u * (unsigned)K) + w * (unsigned)K) + 0.5f * kf);
    P3 pv[2] = {c, c};
    float dv[2];
    uint32_t fired = 0u;
    gsdf_dev::sdf_eval<2, 0, true>(code, pv, dv, lds, BLOCK, false, radius, (uint32_t)lip_base, 0u, &fired);
    const float pad = 4e-6f * (dm::absf(c.x) + dm::absf(c.y) + dm::absf(c.z) + radius);
    const bool none = dv[0] >= keepDist + pad || dv[1] <= -(keepDist + pad);  // (NaN bounds: false, the block is evaluated)
    if (valid) flags[b] = none ? 0u : (0x80000000u | (fired & 0xffffu));
  }
}

// Stage 1 (Reset :26-83): evaluate every cube origin; keep iff |d| < 2*size (octreePrunea szMult=2, origin).
template <int K, int W = 3>
__global__ void __launch_bounds__(BLOCK, W) dc_origin_kernel(const uint32_t* __restrict__ code_g, int nslots, int nshift, float ox, float oy,
                                                             float oz, float res, int* __restrict__ grid, Cube* __restrict__ cubes,
                                                             unsigned long long cube_cap, unsigned zlo, unsigned zhi,
                                                             int use_box, float bx0, float by0, float bz0, float bx1, float by1,
                                                             float bz1, unsigned tx0, unsigned ty0, unsigned tz0, unsigned ntx,
                                                             unsigned nty, unsigned ntz, DCCounters* __restrict__ ctr,
                                                             const uint32_t* __restrict__ block_keep /* dc_block_test_kernel's verdicts and brick masks, or null: every block is evaluated */) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  unsigned* s_w = (unsigned*)(g_smem + (size_t)(nslots > 0 ? nslots : 1) * K * BLOCK);  // 4 wave totals
  unsigned long long* s_base = (unsigned long long*)(s_w + 4);
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // multi-GPU: this rank evaluates the z-slab [zlo, zhi) of the lattice (its own slab plus a one-cube halo).
  // Work tile = a compact 8 x 8 x 4K brick of cells (a wave = a block of 8 x 8 x K of them: an x,y column of K z per lane), not a K*256-long row:
  // spatially coherent waves are what lets D_SKIPFAR* drop the far children of a wide union for the whole wave.
  // The host hands over the range of tiles to sweep: all of the slab, or -- for trees with an exact box -- the tiles
  // that can hold a kept cube or a neighbour of one (box grown by the keep radius and two cells); cells outside that
  // range are neither written here nor read by the later stages.
  const unsigned n = 1u << nshift;
  const float maxDist = res * 2;
  unsigned long long my_evals = 0;
  const uint64_t ntiles = (uint64_t)ntx * nty * ntz;
  // this workgroup's part of the lists (DC_PARTS): its counter word, its share of the cubes' array and of the run descriptors behind it
  const unsigned part = blockIdx.x % DC_PARTS;
  unsigned long long* const my_word = &ctr->cubes_w[part * DC_WORD_STRIDE];
  const unsigned long long cseg = cube_cap / DC_PARTS, cfirst = (unsigned long long)part * cseg;
  unsigned long long* const my_runs = (unsigned long long*)(cubes + cube_cap) + (unsigned long long)part * dc_cube_run_seg(ntiles);
  float* const origin_dist = (float*)((unsigned long long*)(cubes + cube_cap) + dc_origin_dist_offset_words(ntiles));  // [cube]: the edge stage's first point is this one
  for (uint64_t T = blockIdx.x; T < ntiles; T += gridDim.x) {  // block-uniform trip count
    // a tile whose four blocks the interval test cleared: nothing kept, nothing to write (the grid of the range was cleared to -1)
    if (block_keep != nullptr) {
      const uint4 f = *(const uint4*)(block_keep + T * 4ull);  // (block-uniform: a scalar load)
      if (((f.x | f.y | f.z | f.w) >> 31) == 0u) continue;
    }
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
      cz[kp] = zlo + tz * (4u * K) + wave * (unsigned)K + (unsigned)kp;  // (j >> 6 = 4 kp + wave until round 4: the wave's K levels four apart)
      (void)j;
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
    // ... and a wave whose block the interval test cleared (dc_block_test_kernel): nothing of it is within the keep radius
    const uint32_t bflag = block_keep != nullptr ? (uint32_t)__builtin_amdgcn_readfirstlane((int)block_keep[T * 4ull + wave]) : 0x80000000u;
    const bool cleared = (bflag >> 31) == 0u;
    const uint32_t bmask = block_keep != nullptr ? ((bflag & 0xffffu) | GSDF_BRICK_MASK_VALID) : 0u;  // (no test, no mask: the gates are tested per wave)
    if (cleared || __all(far)) {
#pragma unroll
      for (int kp = 0; kp < K; kp++) d[kp] = 3.0e38f;
    } else {
      gsdf_dev::sdf_eval<K, 2>(code, p, d, lds, BLOCK, /*brick=*/true, 0.0f, 0u, bmask);  // COLUMN mode: the lane's K cells are one x,y column (z = K wave + kp: the same in every lane of the wave); a wave = an 8 x 8 patch of columns: polygon edge culling pays (a glyph outline of 30 edges keeps 3-6)
#pragma unroll
      for (int kp = 0; kp < K; kp++) {  // count the lattice cells (tiles overhang the lattice edge)
        const unsigned long long vm = __ballot(valid[kp]);
        if (lane == 0) my_evals += (unsigned long long)__builtin_popcountll(vm);
      }
    }
    // Block-wide append: ONE global atomic per workgroup pass (K*256 cells) instead of one per wave and point.
    bool keep[K];
    unsigned mine = 0;
#pragma unroll
    for (int kp = 0; kp < K; kp++) {
      keep[kp] = valid[kp] && !nb::abs_ge(d[kp], maxDist);
      mine += keep[kp] ? 1u : 0u;
    }
    // Most passes keep nothing (the kept cubes hug the surface, the sweep covers the part's whole box): one barrier tells, and the
    // pass is over -- the scan, two more barriers and the append's round trip are for the passes that have something to append.
    // (profiles/r4_dc_*: the stage issued VALU instructions in 20 % of its slots and waited in 41-49 % of its wave cycles.)
    if (!__syncthreads_or((int)mine)) {  // (also: the previous pass has finished reading s_w / s_base)
#pragma unroll
      for (int kp = 0; kp < K; kp++)
        if (valid[kp]) grid[(uint64_t)cx[kp] + ((uint64_t)cy[kp] << nshift) + ((uint64_t)cz[kp] << (2 * nshift))] = -1;
      continue;
    }
    unsigned incl = mine;  // wave inclusive scan of per-lane counts
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned v = __shfl_up(incl, off, 64);
      if (lane >= (unsigned)off) incl += v;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    const unsigned w0 = s_w[0], w1 = s_w[1], w2 = s_w[2], w3 = s_w[3];
    const unsigned total = w0 + w1 + w2 + w3;  // (> 0 here)
    if (threadIdx.x == 0) {
      const unsigned long long w = atomicAdd(my_word, (unsigned long long)total | (1ull << 36));
#ifdef GSDF_EXP_DC_EXTRA_ATOMIC  // experiment: what a second returning atomic on the same word costs the stage
      if (atomicAdd(my_word, 0ull) == 0x123456789abcull) ctr->t_overflow = 1ull;
#endif
      const unsigned long long b = DC_W_COUNT(w);
      *s_base = b;
      // this pass's run (a run past the part's capacity has no entries: the mesh is repeated with more room)
      const unsigned long long room = b < cseg ? cseg - b : 0ull;
      if (DC_W_RUNS(w) < dc_cube_run_seg(ntiles)) my_runs[DC_W_RUNS(w)] = DC_DESC(cfirst + b, room < total ? room : total);  // (one run per tile of the part at most: the bound holds by construction, the test keeps a wrong launch from writing elsewhere)
      else ctr->q_overflow = 1ull;
    }
    __syncthreads();
    unsigned long long slot = *s_base + (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u) + (incl - mine);  // (within the part)
#pragma unroll
    for (int kp = 0; kp < K; kp++) {
      const uint64_t c = (uint64_t)cx[kp] + ((uint64_t)cy[kp] << nshift) + ((uint64_t)cz[kp] << (2 * nshift));
      if (keep[kp]) {
        if (slot < cseg) {
          cubes[cfirst + slot] = Cube{(uint16_t)cx[kp], (uint16_t)cy[kp], (uint16_t)cz[kp], 0};
          origin_dist[cfirst + slot] = d[kp];
          grid[c] = (int)(cfirst + slot);
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
  {  // statistics: one add per workgroup, spread over 64 words on 64 cache lines
    __syncthreads();
    unsigned long long* s_ev = (unsigned long long*)s_base;  // (the append's word is idle now)
    if (threadIdx.x == 0) *s_ev = 0ull;
    __syncthreads();
    if (lane == 0 && my_evals) atomicAdd(s_ev, my_evals);
    __syncthreads();
    if (threadIdx.x == 0 && *s_ev) atomicAdd(&ctr->n_origin_part[(blockIdx.x & 63u) * 8u], *s_ev);
  }
}

// Stage 2 (RenderAll :85-108): origin, +x, +y, +z distances of every kept cube (one 3-point pass per
// lane + the sweep's value of the origin); default FinalVertex = cube origin; active edges (sign BIT differs, :261-269) are compacted.
// Work item = a run of the origin sweep (DC_DESC): the cubes of one tile, 256 per pass; a wave without a cube of the run
// skips the evaluation, lanes past the run's end repeat its last cube (a dummy position would stretch the wave's bounding box).
// The edges are appended once per run (DC_EDGE_PASSES passes = the 1024 cells of the largest tile: one atomic); until then a pass's active edges wait in
// LDS as 16 bits per lane (which of the cube's three, where they go within the pass's part of the append).
#define DC_EDGE_PASSES 4  // (2 KB of LDS: npt-flange keeps its five workgroups per CU)
__global__ void __launch_bounds__(BLOCK, 3) dc_edges_kernel(const uint32_t* __restrict__ code_g, const Cube* __restrict__ cubes,
                                                            unsigned long long cube_cap, float ox, float oy, float oz, float res,
                                                            float4* __restrict__ dists, float* __restrict__ fv,
                                                            unsigned* __restrict__ edges, unsigned long long edge_cap,
                                                            unsigned long long* __restrict__ edge_runs, unsigned long long ntiles /* of the origin sweep */,
                                                            const int* __restrict__ grid, int nshift, unsigned char* __restrict__ placed /* one per cube, cleared: <- 1 for the cubes around an active edge */,
                                                            DCCounters* __restrict__ ctr) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  __shared__ uint16_t s_pend[DC_EDGE_PASSES][BLOCK];
  __shared__ unsigned s_tot[DC_EDGE_PASSES], s_off[DC_EDGE_PASSES], s_w[4];
  __shared__ unsigned long long s_word;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long* __restrict__ cube_runs = (const unsigned long long*)(cubes + cube_cap);
  const float* __restrict__ origin_dist = (const float*)(cube_runs + dc_origin_dist_offset_words(ntiles));
  const unsigned long long crseg = dc_cube_run_seg(ntiles);
  unsigned long long nr[DC_PARTS];
  const unsigned long long nruns = dc_part_counts<true>(ctr->cubes_w, crseg, nr);  // the cube runs of all parts, one after the other
  // this workgroup's part of the edge list (DC_PARTS): its counter word, its share of the edges' array and of the edge runs
  const unsigned part = blockIdx.x % DC_PARTS;
  unsigned long long* const my_word = &ctr->edges_w[part * DC_WORD_STRIDE];
  const unsigned long long eseg = edge_cap / DC_PARTS, efirst = (unsigned long long)part * eseg;
  unsigned long long* const my_runs = edge_runs + (unsigned long long)part * dc_edge_run_seg(cube_cap, ntiles);
  auto run_at = [&](unsigned long long g) -> unsigned long long {  // (block-uniform)
    unsigned pp = 0;
    unsigned long long kk = 0;
    return dc_flat_to_part(g, nr, pp, kk) ? uniform_u64(cube_runs[(unsigned long long)pp * crseg + kk]) : 0ull;
  };
  unsigned long long dsc_next = run_at(blockIdx.x);
  for (unsigned long long r = blockIdx.x; r < nruns; r += gridDim.x) {  // block-uniform trip counts
    const unsigned long long dsc = dsc_next;
    dsc_next = run_at(r + gridDim.x);  // (its latency under this run's evaluation)
    const unsigned long long first = DC_DESC_FIRST(dsc);
    const unsigned cnt = DC_DESC_COUNT(dsc);  // (0: a run past the cubes' capacity -- the mesh is repeated with more room)
    // Which wave takes which 64 of a pass's 256 changes from run to run: a run's last pass fills one or two waves, and with a fixed
    // order those would always be waves 0 and 1.
    const unsigned rot = (unsigned)__builtin_popcountll(r) & 3u;
    const unsigned jt = (((threadIdx.x >> 6) - rot) & 3u) * 64u + lane;  // this thread's place in a pass
    unsigned npend = 0;  // block-uniform
    for (unsigned off = 0; off < cnt; off += BLOCK) {
      const unsigned j = off + jt;
      const bool valid = j < cnt;
      const uint64_t i = first + (valid ? j : cnt - 1u);
      const Cube c = cubes[i];
      const float x0 = ox + res * (float)c.x, y0 = oy + res * (float)c.y, z0 = oz + res * (float)c.z;
      // The cube's own origin was evaluated by the sweep -- the same float position (O + res * i, formed the same way), hence the same
      // bits: it is read back; the three points a cell further along x, y, z are evaluated here (the reference forms them as
      // origin + res, not as the neighbour's O + res * (i + 1): they are not the neighbours' origins bit for bit).
      P3 p[3] = {{x0 + res, y0 + 0.f, z0 + 0.f}, {x0 + 0.f, y0 + res, z0 + 0.f}, {x0 + 0.f, y0 + 0.f, z0 + res}};
      float d3[3] = {0.f, 0.f, 0.f};
      if (__builtin_amdgcn_ballot_w64(valid) != 0ull)  // (wave-uniform)
        gsdf_dev::sdf_eval<3>(code, p, d3, lds, BLOCK, /*brick=*/true);  // (a wave's cubes are neighbours within one tile: edge culling on)
      const float d[4] = {origin_dist[i], d3[0], d3[1], d3[2]};
      if (valid) {
        dists[i] = make_float4(d[0], d[1], d[2], d[3]);
        fv[3 * i] = x0; fv[3 * i + 1] = y0; fv[3 * i + 2] = z0;
      }
      const unsigned s0 = __float_as_uint(d[0]) >> 31;
      unsigned act = 0, mine = 0;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const bool on = valid && ((__float_as_uint(d[1 + a]) >> 31) != s0);
        act |= on ? (1u << a) : 0u;
        mine += on ? 1u : 0u;
      }
      unsigned incl = mine;  // where this pass's edges go among themselves: a block-wide scan (thread order)
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(incl, o, 64);
        if (lane >= (unsigned)o) incl += v;
      }
      __syncthreads();  // the previous pass's (or flush's) readers are done with s_w, s_pend, s_tot
      if (lane == 63u) s_w[wave] = incl;
      __syncthreads();
      const unsigned w0 = s_w[0], w1 = s_w[1], w2 = s_w[2], w3 = s_w[3];
      s_pend[npend][threadIdx.x] = (uint16_t)(act | (((wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u) + (incl - mine)) << 3));  // (< 768 ahead of it)
      if (threadIdx.x == 0) { s_tot[npend] = w0 + w1 + w2 + w3; s_off[npend] = off; }
      npend++;
      if (npend < DC_EDGE_PASSES && off + BLOCK < cnt) continue;
      // ---- the append of the passes that wait: one atomic for their edges and the runs of <= 64 they are handed to the normals stage in
      __syncthreads();
      unsigned sum = 0;
      for (unsigned q = 0; q < npend; q++) sum += s_tot[q];
      if (sum != 0u) {  // (block-uniform)
        if (threadIdx.x == 0) s_word = atomicAdd(my_word, (unsigned long long)sum | ((unsigned long long)((sum + 63u) >> 6) << 36));
#ifdef GSDF_EXP_DC_EXTRA_ATOMIC
        if (threadIdx.x == 0 && atomicAdd(my_word, 0ull) == 0x123456789abcull) ctr->t_overflow = 1ull;
#endif
        __syncthreads();
        const unsigned long long wd = s_word;
        const unsigned long long ebase = DC_W_COUNT(wd);  // (within the part)
        unsigned long long base = ebase;
        for (unsigned q = 0; q < npend; q++) {
          const unsigned pend = s_pend[q][threadIdx.x];
          const unsigned cube = (unsigned)(first + s_off[q] + jt);
          unsigned long long slot = base + (pend >> 3);
#pragma unroll
          for (int a = 0; a < 3; a++) {
            if ((pend >> a) & 1u) {
              if (slot < eseg) edges[efirst + slot] = (cube << 2) | (unsigned)a;
              else ctr->q_overflow = 1ull;
              slot++;
            }
          }
          base += s_tot[q];
          // The cubes this cube's active edges belong to -- itself and, per edge, the three at -1 along the other two axes
          // (EdgeNeighborsX/Y/Z, dual_contour.go:271-287) -- are the ones the placement stage solves for (len(cube.Neighbors) != 0,
          // dual_contour_vertexplacement.go:52-60): marked here, three grid lookups per active edge, instead of every kept cube
          // walking the eight cubes of its cell there (22 M lookups against 2.5 M for npt-flange at resdiv 800).
          if ((pend & 7u) != 0u) {
            const Cube c = cubes[cube];
            placed[cube] = 1;
            const int nn = 1 << nshift;
            (void)nn;
#pragma unroll
            for (int a = 0; a < 3; a++) {
              if ((pend >> a) & 1u) {
                const int b1 = (a + 1) % 3, b2 = (a + 2) % 3;
#pragma unroll
                for (int k = 1; k < 4; k++) {
                  int o[3] = {0, 0, 0};
                  o[b1] = -(k & 1); o[b2] = -(k >> 1);
                  const int x = (int)c.x + o[0], y = (int)c.y + o[1], z = (int)c.z + o[2];
                  if (x < 0 || y < 0 || z < 0) continue;
                  const int ui = grid[((size_t)z << (2 * nshift)) + ((size_t)y << nshift) + (size_t)x];
                  if (ui >= 0) placed[ui] = 1;
                }
              }
            }
          }
        }
        const unsigned long long room = ebase < eseg ? eseg - ebase : 0ull;
        const unsigned n = room < sum ? (unsigned)room : sum, nsub = (sum + 63u) >> 6;
        const unsigned long long erseg = dc_edge_run_seg(cube_cap, ntiles);
        for (unsigned k = threadIdx.x; k < nsub; k += BLOCK) {
          // (a part whose edges overflow its share of the array -- the mesh is repeated then -- may also run out of descriptors)
          if (DC_W_RUNS(wd) + k < erseg) my_runs[DC_W_RUNS(wd) + k] = DC_DESC(efirst + ebase + 64u * k, 64u * k >= n ? 0u : (n - 64u * k < 64u ? n - 64u * k : 64u));
          else ctr->q_overflow = 1ull;
        }
      }
      npend = 0;
    }
  }
}

__device__ __forceinline__ float dc_isect(float o, float e) { return -o / (e - o); }

// Stage 3 (PlaceVertices :28-50 + gleval.NormalsCentralDiff): raw central-difference normals at the
// linear intersection of every ACTIVE edge (inactive edges' normals are never read by the reference).
// Work item = a wave-sized run of edges from one pass of the edge stage (DC_DESC), a wave each: no workgroup-level step here.
__global__ void __launch_bounds__(BLOCK, 3) dc_normals_kernel(const uint32_t* __restrict__ code_g, const Cube* __restrict__ cubes,
                                                              const float4* __restrict__ dists, const unsigned* __restrict__ edges,
                                                              const unsigned long long* __restrict__ edge_runs, unsigned long long cube_cap,
                                                              unsigned long long ntiles, float ox, float oy, float oz,
                                                              float res, float h, float* __restrict__ nrm, DCCounters* __restrict__ ctr) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  const unsigned long long erseg = dc_edge_run_seg(cube_cap, ntiles);
  unsigned long long nr[DC_PARTS];
  const unsigned long long nruns = dc_part_counts<true>(ctr->edges_w, erseg, nr);  // the edge runs of all parts, one after the other
  const unsigned lane = threadIdx.x & 63u;
  for (unsigned long long r = uniform_u64((unsigned long long)blockIdx.x * 4u + (threadIdx.x >> 6)); r < nruns; r += (unsigned long long)gridDim.x * 4u) {  // wave-uniform
    unsigned pp = 0;
    unsigned long long kk = 0;
    dc_flat_to_part(r, nr, pp, kk);
    const unsigned long long dsc = uniform_u64(edge_runs[(unsigned long long)pp * erseg + kk]);
    const unsigned cnt = DC_DESC_COUNT(dsc);  // 1 .. 64 (0: past the edges' capacity)
    if (cnt == 0u) continue;
    const bool valid = lane < cnt;
    const unsigned e = edges[DC_DESC_FIRST(dsc) + (valid ? lane : cnt - 1u)];  // (lanes past the end repeat the last edge: see dc_edges_kernel)
    const unsigned ci = e >> 2, a = e & 3u;
    const Cube c = cubes[ci];
    const float4 d = dists[ci];
    const float t = res * dc_isect(d.x, a == 0 ? d.y : (a == 1 ? d.z : d.w));
    const float px = (ox + res * (float)c.x) + (a == 0 ? t : 0.f);
    const float py = (oy + res * (float)c.y) + (a == 1 ? t : 0.f);
    const float pz = (oz + res * (float)c.z) + (a == 2 ? t : 0.f);
    float out[3];
#pragma unroll 1
    for (int dim = 0; dim < 3; dim++) {
      P3 ab[2] = {{px + (dim == 0 ? h : 0.f), py + (dim == 1 ? h : 0.f), pz + (dim == 2 ? h : 0.f)},
                  {px - (dim == 0 ? h : 0.f), py - (dim == 1 ? h : 0.f), pz - (dim == 2 ? h : 0.f)}};
      float dd[2];
      gsdf_dev::sdf_eval<2>(code, ab, dd, lds, BLOCK, /*brick=*/true);  // (the wave's edges lie within one tile: edge culling on. Round 4, waves that straddled runs from different places: 0.255 -> 0.31 ms with it)
      const float v = dd[0] - dd[1];
      if (dim == 0) out[0] = v; else if (dim == 1) out[1] = v; else out[2] = v;
    }
    if (valid) {
      const size_t o = ((size_t)ci * 3 + a) * 3;
      nrm[o] = out[0]; nrm[o + 1] = out[1]; nrm[o + 2] = out[2];
    }
  }
}

#define DC_ROWS 18  // <= 3 own + 12 contributed (own edges appear again among them) + 3 regularisation rows
#define DC_BLOCK 64
// Stage 4 (PlaceVertices :52-141, leastSquaresMGS64 :152-223): per cube, rows = own active edges, then the
// edges of the (up to 12) contributing cubes in lattice order (z,y,x) and axis order, 3 regularisation rows;
// float64 modified Gram-Schmidt. Every entry of the system is a float32 value (normals, products formed in float32), so the rows
// are KEPT as float32 and the orthonormalised columns the reference keeps in place (A[k][j] -= dot * A[k][i]; A[k][j] *= inv)
// are recomputed from them wherever they are read: the same float64 operations on the same operands in the same order, hence
// the same bits. Since round 5 the rows are in registers at fixed places (see place() below; until then in LDS, [row][col][lane],
// 18.4 KB per 64 lanes).
__global__ void __launch_bounds__(DC_BLOCK) dc_place_kernel(const Cube* __restrict__ cubes, unsigned long long cube_cap,
                                                            const float4* __restrict__ dists, const int* __restrict__ grid,
                                                            const float* __restrict__ nrm, int nshift, float ox, float oy, float oz,
                                                            float res, float sqrtLambda, float* __restrict__ fv, unsigned zplace_hi,
                                                            const unsigned char* __restrict__ placed_flag /* dc_edges_kernel's marks */, DCCounters* __restrict__ ctr) {
  const unsigned long long cseg = cube_cap / DC_PARTS;
  unsigned long long np[DC_PARTS];
  const unsigned long long n = dc_part_counts<false>(ctr->cubes_w, cseg, np);  // the cubes of all parts, one after the other
  const int nn = 1 << nshift;
  const unsigned t = threadIdx.x;
  // Two cubes in five are placed (the ones with an active edge among the twelve of their cell; the kept cubes are a band four cells
  // thick around the surface), and taken 64 at a time in list order the least-squares solve ran with that share of its lanes. So the
  // workgroup -- one wave -- first asks which of its next 64 cubes are placed (a byte per cube, set by the edge stage) and queues those
  // (a ring of 128 indices in LDS); the solve runs whenever 64 are waiting, on full waves.
  __shared__ unsigned s_q[128];
  unsigned head = 0, tail = 0;  // (wave-uniform)
  auto place = [&](const uint64_t i) {
    const Cube c = cubes[i];
    const float cox = ox + res * (float)c.x, coy = oy + res * (float)c.y, coz = oz + res * (float)c.z;
    const float invRes = 1.0f / res;
    // The system's rows live in REGISTERS, one fixed place per edge of the cell: 0..2 the cube's own edges, 3..14 the cell's twelve in
    // lattice order (z, y, x) and axis order -- the order the reference appends them in -- 15..17 the regularisation rows; an edge
    // that is not active leaves its row zero, and a zero row adds +-0 to sums that start at +0: an exact no-op (the sums never
    // hold -0: they start at +0 and a cancellation gives +0). Every loop over the rows is unrolled: no LDS, no row count in the
    // loops' bounds -- the kernel had 18.4 KB of LDS per wave (two waves per SIMD) and a trip to LDS for every operand.
    float A0[DC_ROWS], A1[DC_ROWS], A2[DC_ROWS], Bv[DC_ROWS];
#pragma unroll
    for (int k = 0; k < DC_ROWS; k++) { A0[k] = 0.f; A1[k] = 0.f; A2[k] = 0.f; Bv[k] = 0.f; }
    int nr = 0;
    float mx = 0.f, my = 0.f, mz = 0.f;
    auto edge_row = [&](const int slot, unsigned ci, int a) {  // (slot: a constant after unrolling)
      const Cube u = cubes[ci];
      const float4 d = dists[ci];
      const float tt = res * dc_isect(d.x, a == 0 ? d.y : (a == 1 ? d.z : d.w));
      const float ux = ox + res * (float)u.x, uy = oy + res * (float)u.y, uz = oz + res * (float)u.z;
      const size_t o = ((size_t)ci * 3 + (size_t)a) * 3;
      const float bx = ux + (a == 0 ? tt : 0.f), by = uy + (a == 1 ? tt : 0.f), bz = uz + (a == 2 ? tt : 0.f);
      const float nx = nrm[o], ny = nrm[o + 1], nz = nrm[o + 2];
      const float qx = invRes * (bx - cox), qy = invRes * (by - coy), qz = invRes * (bz - coz);
      A0[slot] = nx; A1[slot] = ny; A2[slot] = nz;
      Bv[slot] = nx * qx + ny * qy + nz * qz;
      mx = mx + bx; my = my + by; mz = mz + bz;
      nr++;
    };
    // (whether the cube is placed at all -- len(cube.Neighbors) != 0 -- was decided when it was queued)
    {
      const float4 d = dists[i];
      const unsigned s0 = __float_as_uint(d.x) >> 31;
      if ((__float_as_uint(d.y) >> 31) != s0) edge_row(0, (unsigned)i, 0);
      if ((__float_as_uint(d.z) >> 31) != s0) edge_row(1, (unsigned)i, 1);
      if ((__float_as_uint(d.w) >> 31) != s0) edge_row(2, (unsigned)i, 2);
    }
    // the active edges of the cell, in lattice order (z, y, x) and axis order
    {
      int slot = 3;
#pragma unroll
      for (int dz = 0; dz < 2; dz++)
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
          for (int dx = 0; dx < 2; dx++) {
            const int s_x = slot, s_y = slot + (dx == 0 ? 1 : 0), s_z = s_y + (dy == 0 ? 1 : 0);  // this neighbour's places: x, y, z edges as far as it has them
            slot = s_z + (dz == 0 ? 1 : 0);
            const int ux = c.x + dx, uy = c.y + dy, uz = c.z + dz;
            if (ux >= nn || uy >= nn || uz >= nn) continue;
            const int ui = grid[((size_t)uz * nn + uy) * nn + ux];
            if (ui < 0) continue;
            const float4 d = dists[ui];
            const unsigned s0 = __float_as_uint(d.x) >> 31;
            if (dx == 0 && ((__float_as_uint(d.y) >> 31) != s0)) edge_row(s_x, (unsigned)ui, 0);
            if (dy == 0 && ((__float_as_uint(d.z) >> 31) != s0)) edge_row(s_y, (unsigned)ui, 1);
            if (dz == 0 && ((__float_as_uint(d.w) >> 31) != s0)) edge_row(s_z, (unsigned)ui, 2);
          }
    }
    const float im = 1.f / (float)nr;
    const float bsx = invRes * (im * mx - cox), bsy = invRes * (im * my - coy), bsz = invRes * (im * mz - coz);
    A0[15] = sqrtLambda; Bv[15] = sqrtLambda * bsx;
    A1[16] = sqrtLambda; Bv[16] = sqrtLambda * bsy;
    A2[17] = sqrtLambda; Bv[17] = sqrtLambda * bsz;
    double R[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    // leastSquaresMGS64 :152-223, column by column; qJ(k) = what the reference's A[k][J] holds once column J is done
    // column 0: normalise
    double nsq = 0;
#pragma unroll
    for (int k = 0; k < DC_ROWS; k++) { const double a = (double)A0[k]; nsq += a * a; }
    const double norm0 = __builtin_sqrt(nsq);
    R[0][0] = norm0;
    const bool have0 = norm0 > 1e-14;
    const double inv0 = 1.0 / norm0;
#define DC_Q0(k) (have0 ? (double)A0[k] * inv0 : (double)A0[k])
    // column 1: minus its projection on q0, then normalise
    double dot01 = 0;
#pragma unroll
    for (int k = 0; k < DC_ROWS; k++) dot01 += DC_Q0(k) * (double)A1[k];
    R[0][1] = dot01;
#define DC_V1(k) ((double)A1[k] - dot01 * DC_Q0(k))
    nsq = 0;
#pragma unroll
    for (int k = 0; k < DC_ROWS; k++) { const double v = DC_V1(k); nsq += v * v; }
    const double norm1 = __builtin_sqrt(nsq);
    R[1][1] = norm1;
    const bool have1 = norm1 > 1e-14;
    const double inv1 = 1.0 / norm1;
#define DC_Q1(k) (have1 ? DC_V1(k) * inv1 : DC_V1(k))
    // column 2: minus its projections on q0 and (what is left) on q1, then normalise
    double dot02 = 0;
#pragma unroll
    for (int k = 0; k < DC_ROWS; k++) dot02 += DC_Q0(k) * (double)A2[k];
    R[0][2] = dot02;
#define DC_W2(k) ((double)A2[k] - dot02 * DC_Q0(k))
    double dot12 = 0;
#pragma unroll
    for (int k = 0; k < DC_ROWS; k++) dot12 += DC_Q1(k) * DC_W2(k);
    R[1][2] = dot12;
#define DC_V2(k) (DC_W2(k) - dot12 * DC_Q1(k))
    nsq = 0;
#pragma unroll
    for (int k = 0; k < DC_ROWS; k++) { const double v = DC_V2(k); nsq += v * v; }
    const double norm2 = __builtin_sqrt(nsq);
    R[2][2] = norm2;
    const bool have2 = norm2 > 1e-14;
    const double inv2 = 1.0 / norm2;
    double Qtb[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < DC_ROWS; k++) Qtb[0] += DC_Q0(k) * (double)Bv[k];
#pragma unroll
    for (int k = 0; k < DC_ROWS; k++) Qtb[1] += DC_Q1(k) * (double)Bv[k];
#pragma unroll
    for (int k = 0; k < DC_ROWS; k++) { const double v = DC_V2(k); Qtb[2] += (have2 ? v * inv2 : v) * (double)Bv[k]; }
#undef DC_Q0
#undef DC_V1
#undef DC_Q1
#undef DC_W2
#undef DC_V2
    double x[3];
    for (int ii = 2; ii >= 0; ii--) {
      x[ii] = Qtb[ii];
      for (int k = ii + 1; k < 3; k++) x[ii] -= R[ii][k] * x[k];
      if (R[ii][ii] > 1e-14) x[ii] /= R[ii][ii];
      else x[ii] = 0;
    }
    const float xf = dm::clampf((float)x[0], -0.1f, 1.1f), yf = dm::clampf((float)x[1], -0.1f, 1.1f), zf = dm::clampf((float)x[2], -0.1f, 1.1f);
    fv[3 * i] = res * xf + cox; fv[3 * i + 1] = res * yf + coy; fv[3 * i + 2] = res * zf + coz;
  };
  auto drain = [&](unsigned count) {  // the cubes at the ring's head, a lane each (each lane owns column t of the LDS arrays)
    __syncthreads();  // (one wave: the queue's writes are in LDS)
    if (t < count) place((uint64_t)s_q[(head + t) & 127u]);
    head += count;
  };
  const uint64_t step = (uint64_t)gridDim.x * DC_BLOCK;
  for (uint64_t base = (uint64_t)blockIdx.x * DC_BLOCK;; base += step) {  // wave-uniform trip count; past the list's end: what is left in the queue
    const bool more = base < n;
    unsigned pp = 0;
    unsigned long long kk = 0;
    bool placed = false;
    unsigned idx = 0;
    if (more && dc_flat_to_part(base + t, np, pp, kk)) {
      const uint64_t i = (uint64_t)pp * cseg + kk;
      idx = (unsigned)i;
      // len(cube.Neighbors) != 0: an active edge among the cell's twelve -- the edge stage marked those cubes (top halo layer of a
      // z-slab: only its distances / normals are needed)
      placed = placed_flag[i] != 0 && cubes[i].z < zplace_hi;
    }
    const unsigned long long m = __ballot(placed);
    if (placed) s_q[(tail + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))) & 127u] = idx;
    tail += (unsigned)__builtin_popcountll(m);
    const unsigned waiting = tail - head;
    if (waiting >= 64u || (!more && waiting != 0u)) drain(waiting < 64u ? waiting : 64u);  // (the solve's only call site: it is inlined once)
    if (!more && tail == head) break;
  }
}

// Stage 5 (RenderAll :143-219): one quad (2 triangles) per active edge whose 4 surrounding cubes exist.
__global__ void __launch_bounds__(BLOCK) dc_quads_kernel(const Cube* __restrict__ cubes, const float4* __restrict__ dists,
                                                         const unsigned* __restrict__ edges, unsigned long long edge_cap,
                                                         const int* __restrict__ grid, const float* __restrict__ fv, int nshift,
                                                         unsigned zown_lo, unsigned zown_hi, float* __restrict__ tris,
                                                         unsigned long long tri_cap, DCCounters* __restrict__ ctr) {
  const unsigned long long eseg = edge_cap / DC_PARTS;
  unsigned long long np[DC_PARTS];
  const unsigned long long n = dc_part_counts<false>(ctr->edges_w, eseg, np);  // the edges of all parts, one after the other
  const int nn = 1 << nshift;
  const uint64_t step = (uint64_t)gridDim.x * BLOCK;
  for (uint64_t base = (uint64_t)blockIdx.x * BLOCK; base < n; base += step) {
    unsigned pp = 0;
    unsigned long long kk = 0;
    bool ok = dc_flat_to_part(base + threadIdx.x, np, pp, kk);
    int q[4] = {-1, -1, -1, -1};
    bool flip = false;
    if (ok) {
      const unsigned e = edges[(uint64_t)pp * eseg + kk];
      const unsigned ci = e >> 2, a = e & 3u;
      const Cube c = cubes[ci];
      const float4 d = dists[ci];
      ok = ok && c.z >= zown_lo && c.z < zown_hi;  // quads are emitted by the rank that owns the edge's cube
      flip = nb::lt0((a == 0 ? d.y : (a == 1 ? d.z : d.w)) - d.x);
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
    const unsigned long long slot = block_append_n(ok ? 1u : 0u, &ctr->n_tris);  // one atomic per workgroup pass
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
