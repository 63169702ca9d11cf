// kernels_flat.h -- glrender.FlatRenderer on device (glrender/flatrenderer.go:36-256): the lattice pass and the marching passes.
#pragma once
#include "kernels_octree.h"  // marching-cubes emission helpers, record layout

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
      const unsigned long long ng = __ballot(ok && nb::lt0(d[z])), nr = __ballot(ok && nb::abs_le(d[z], cubeDiag));
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
      any_act = any_act || (cube_x && cy0 + (unsigned)r < ny && nb::abs_le(d0[r], cubeDiag));
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
      const bool act = cube_x && cy0 + (unsigned)r < ny && nb::abs_le(d0[r], cubeDiag);  // the reference's |d0| <= 2*sqrt3*res test (:207-209)
      const unsigned long long am = __ballot(act);
      if (am == 0ull) continue;  // wave-uniform
      my_active += (unsigned)__builtin_popcountll(am);
      const float c3 = r + 1 < FLAT_ROWS ? d0[r + 1 < FLAT_ROWS ? r + 1 : 0] : d8;
      const float c1 = __shfl_down(d0[r], 1, 64), c2 = __shfl_down(c3, 1, 64);
      const float c4 = up[r], c7 = up[r + 1], c5 = __shfl_down(c4, 1, 64), c6 = __shfl_down(c7, 1, 64);
      unsigned ix = 0;
      if (act) {
        ix = (nb::lt0(d0[r]) ? 1u : 0u) | (nb::lt0(c1) ? 2u : 0u) | (nb::lt0(c2) ? 4u : 0u) | (nb::lt0(c3) ? 8u : 0u) | (nb::lt0(c4) ? 16u : 0u) |
             (nb::lt0(c5) ? 32u : 0u) | (nb::lt0(c6) ? 64u : 0u) | (nb::lt0(c7) ? 128u : 0u);
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
