// Micro-benchmark: scalar-fp32 VALU issue rate on gfx950 as a function of waves per SIMD and of the number of
// independent dependency chains per wave (what does the interpreter need to saturate a SIMD?).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  float s[CHAINS];
  for (int i = 0; i < CHAINS; i++) s[i] = a * (float)(threadIdx.x + i);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 32 / CHAINS; r++) {
#pragma unroll
      for (int i = 0; i < CHAINS; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));
    }
  }
  float r = 0;
  for (int i = 0; i < CHAINS; i++) r += s[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
  hipDeviceProp_t pr;
  (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount, iters = 20000;
  float* d;
  (void)hipMalloc(&d, cus * 8 * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto run = [&](int chains, int wps, auto kern) {
    const int grid = cus * wps;  // one 256-thread workgroup = one wave on each of the 4 SIMDs
    for (int rep = 0; rep < 2; rep++) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
    }
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double per_simd = (double)wps * iters * 32;
    printf("chains %2d waves/SIMD %d : %7.3f ms  %5.2f cycles@2.4GHz per wave-instr per SIMD  (%.1f T lane-ops/s)\n", chains, wps, ms,
           ms * 1e6 / per_simd * 2.4, (double)cus * 4 * per_simd * 64 / (ms * 1e-3) / 1e12);
  };
  for (int wps : {1, 2, 3, 4, 8}) {
    run(1, wps, k<1>);
    run(2, wps, k<2>);
    run(4, wps, k<4>);
    run(8, wps, k<8>);
    run(16, wps, k<16>);
  }
  return 0;
}
