// kernels_stl.h -- glrender.WriteBinarySTL on device (glrender/stl.go:15-62).
#pragma once
#include "kernels_common.h"

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
