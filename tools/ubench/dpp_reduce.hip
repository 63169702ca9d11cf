// Check of interp.h's wave_minmax (DPP row_shr / row_bcast wave64 reduction) against a __shfl_xor butterfly:
//   hipcc --offload-arch=gfx950 -O3 -I gsdf_amd/csrc -I include tools/ubench/dpp_reduce.hip -o /tmp/dpp_reduce && /tmp/dpp_reduce
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "interp.h"

__global__ void k(const float* in, unsigned long long* bad) {
  const float v = in[blockIdx.x * 64 + threadIdx.x];
  float mn = v, mx = v;
  for (int m = 1; m < 64; m <<= 1) {
    mn = fminf(mn, __shfl_xor(mn, m, 64));
    mx = fmaxf(mx, __shfl_xor(mx, m, 64));
  }
  const float a = gsdf_dev::wave_minmax<false>(v), b = gsdf_dev::wave_minmax<true>(v);
  if (__float_as_uint(a) != __float_as_uint(mn) || __float_as_uint(b) != __float_as_uint(mx)) atomicAdd(bad, 1ull);
}

int main() {
  const int waves = 1 << 16;
  std::vector<float> h((size_t)waves * 64);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < h.size(); i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const uint32_t r = (uint32_t)(s >> 20);
    float f;
    switch ((s >> 60) & 7) {
      case 0: f = (float)(int32_t)r * 1e-3f; break;
      case 1: f = -(float)(r & 0xffff); break;
      case 2: f = (r & 1) ? __builtin_inff() : -__builtin_inff(); break;
      case 3: f = 0.0f; break;
      default: { uint32_t u = (r & 0x7f7fffffu) | (r & 0x80000000u); __builtin_memcpy(&f, &u, 4); }  // any finite float
    }
    // the extreme of a wave in a chosen lane, so that every lane position is exercised as the source
    h[i] = f;
  }
  for (int w = 0; w < waves; w++) { h[(size_t)w * 64 + (w & 63)] = -3.0e38f; h[(size_t)w * 64 + ((w >> 6) & 63)] = (w & 64) ? 3.0e38f : h[(size_t)w * 64 + ((w >> 6) & 63)]; }
  float* d; unsigned long long *db, hb = 0;
  hipMalloc(&d, h.size() * 4); hipMalloc(&db, 8);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemset(db, 0, 8);
  hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, d, db);
  hipMemcpy(&hb, db, 8, hipMemcpyDeviceToHost);
  printf("waves %d mismatching lanes %llu\n", waves, hb);
  return hb != 0;
}
