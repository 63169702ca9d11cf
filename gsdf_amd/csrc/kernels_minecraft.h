// kernels_minecraft.h -- minecraftRender (glrender/dual_contour.go:297-403): every level-1 cube of the top cube gets its origin and
// the three edge ends (+x, +y, +z) evaluated; a sign change along an edge puts the square face that crosses it into the mesh (two
// triangles, wound by which end is inside). Unexported and test-only in the reference (glrender_test.go:55-81); kept small here:
// positions to HBM, the program's ordinary Evaluate kernel over them, one pass that counts or emits the faces. Not part of the
// specialiser's sources (the evaluation goes through eval_kernel, interpreter or per-tree build alike).
#pragma once
#include "kernels_common.h"

// cube c (x fastest) -> its four positions, in the reference's order: origin, +x, +y, +z (dual_contour.go:323-335; ms3.Add adds
// the zero components too: y + 0 turns a -0 into +0, kept)
__global__ void __launch_bounds__(BLOCK) mcr_positions_kernel(float ox, float oy, float oz, float res, int nshift, uint64_t n_cubes, float* __restrict__ pos) {
  const uint64_t step = (uint64_t)gridDim.x * BLOCK;
  const unsigned mask = (1u << nshift) - 1u;
  for (uint64_t c = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; c < n_cubes; c += step) {
    const unsigned x = (unsigned)c & mask, y = (unsigned)(c >> nshift) & mask, z = (unsigned)(c >> (2 * nshift));
    const float sz = res;  // CubeSize of a level-1 cube
    const float sx = ox + sz * (float)x, sy = oy + sz * (float)y, sq = oz + sz * (float)z;  // CubeOrigin
    float* q = pos + c * 12;
    q[0] = sx;      q[1] = sy;        q[2] = sq;
    q[3] = sx + sz; q[4] = sy + 0.0f; q[5] = sq + 0.0f;
    q[6] = sx + 0.0f; q[7] = sy + sz; q[8] = sq + 0.0f;
    q[9] = sx + 0.0f; q[10] = sy + 0.0f; q[11] = sq + sz;
  }
}

__device__ __forceinline__ void mcr_put(float* t, float ax, float ay, float az, float bx, float by, float bz, float cx, float cy, float cz, bool flip) {
  // a flipped face has its first and third vertex exchanged (dual_contour.go:355-358)
  t[0] = flip ? cx : ax; t[1] = flip ? cy : ay; t[2] = flip ? cz : az;
  t[3] = bx; t[4] = by; t[5] = bz;
  t[6] = flip ? ax : cx; t[7] = flip ? ay : cy; t[8] = flip ? az : cz;
}

// tris == nullptr: count only (ctr[0] += 2 per active edge); else emit at the slots an atomic hands out (any order: the mesh is a set)
__global__ void __launch_bounds__(BLOCK) mcr_faces_kernel(const float* __restrict__ dist, float ox, float oy, float oz, float res, int nshift,
                                                          uint64_t n_cubes, float* __restrict__ tris, unsigned long long cap, unsigned long long* __restrict__ ctr) {
  const uint64_t step = (uint64_t)gridDim.x * BLOCK;
  const unsigned mask = (1u << nshift) - 1u;
  for (uint64_t c = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; c < n_cubes; c += step) {
    const float d0 = dist[4 * c], dx = dist[4 * c + 1], dy = dist[4 * c + 2], dz = dist[4 * c + 3];
    const unsigned s0 = __float_as_uint(d0) >> 31;
    const bool ax = s0 != (__float_as_uint(dx) >> 31), ay = s0 != (__float_as_uint(dy) >> 31), az = s0 != (__float_as_uint(dz) >> 31);  // ActiveX/Y/Z
    const unsigned nt = 2u * ((ax ? 1u : 0u) + (ay ? 1u : 0u) + (az ? 1u : 0u));
    if (nt == 0u) continue;
    unsigned long long at = atomicAdd(ctr, (unsigned long long)nt);
    if (tris == nullptr || at + nt > cap) continue;
    const unsigned x = (unsigned)c & mask, y = (unsigned)(c >> nshift) & mask, z = (unsigned)(c >> (2 * nshift));
    const float sz = res;
    const float sx = ox + sz * (float)x, sy = oy + sz * (float)y, sq = oz + sz * (float)z;
    float* t = tris + at * 9;
    if (ax) {
      const float px = sx + sz, py = sy + 0.0f, pz = sq + 0.0f;  // xOrig
      const bool flip = nb::lt0(dx - d0);                        // FlipX: XDist - OrigDist < 0
      mcr_put(t, px, py, pz, px + 0.0f, py + sz, pz + 0.0f, px + 0.0f, py + sz, pz + sz, flip);
      mcr_put(t + 9, px + 0.0f, py + sz, pz + sz, px + 0.0f, py + 0.0f, pz + sz, px, py, pz, flip);
      t += 18;
    }
    if (ay) {
      const float px = sx + 0.0f, py = sy + sz, pz = sq + 0.0f;  // yOrig
      const bool flip = nb::lt0(dy - d0);
      mcr_put(t, px, py, pz, px + 0.0f, py + 0.0f, pz + sz, px + sz, py + 0.0f, pz + sz, flip);
      mcr_put(t + 9, px + sz, py + 0.0f, pz + sz, px + sz, py + 0.0f, pz + 0.0f, px, py, pz, flip);
      t += 18;
    }
    if (az) {
      const float px = sx + 0.0f, py = sy + 0.0f, pz = sq + sz;  // zOrig
      const bool flip = nb::lt0(dz - d0);
      mcr_put(t, px, py, pz, px + sz, py + 0.0f, pz + 0.0f, px + sz, py + sz, pz + 0.0f, flip);
      mcr_put(t + 9, px + sz, py + sz, pz + 0.0f, px + 0.0f, py + sz, pz + 0.0f, px, py, pz, flip);
    }
  }
}
