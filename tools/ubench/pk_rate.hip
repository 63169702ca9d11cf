// Micro-benchmark: issue rate of v_pk_{fma,mul,add}_f32 against the scalar forms on gfx950 (one workgroup of 256
// threads per CU x 3, 8 independent accumulator chains per lane). Prints cycles-equivalent ns per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  float s[16];
  v2f p[8];
  for (int i = 0; i < 16; i++) s[i] = a * (float)(threadIdx.x + i);
  for (int i = 0; i < 8; i++) p[i] = v2f{s[2 * i], s[2 * i + 1]};
  v2f bb = v2f{b, b}, aa = v2f{a, a};
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {  // 16 scalar fma
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));
    } else if (MODE == 1) {  // 8 packed fma (= 16 fp32 fma)
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(aa), "v"(bb));
    } else if (MODE == 2) {  // 16 scalar mul
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(a));
    } else if (MODE == 3) {  // 8 packed mul
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(aa));
    } else if (MODE == 4) {  // 16 scalar add
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(b));
    } else if (MODE == 5) {  // 8 packed add
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(bb));
    } else if (MODE == 6) {  // 16 f64 fma (8 chains x2)
      // handled below
    }
  }
  float r = 0;
  for (int i = 0; i < 16; i++) r += s[i];
  for (int i = 0; i < 8; i++) r += p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

__global__ void __launch_bounds__(256) k64(double* out, int iters, double a, double b) {
  double s[8];
  for (int i = 0; i < 8; i++) s[i] = a * (double)(threadIdx.x + i);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));
  }
  double r = 0;
  for (int i = 0; i < 8; i++) r += s[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
__global__ void __launch_bounds__(256) ktrans(float* out, int iters, float a) {
  float s[8];
  for (int i = 0; i < 8; i++) s[i] = a * (float)(threadIdx.x + i + 1);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == 0) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[i]));
      if (MODE == 1) asm volatile("v_sqrt_f32 %0, %0" : "+v"(s[i]));
      if (MODE == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(a));
      if (MODE == 3) asm volatile("v_min_f32 %0, %0, %1" : "+v"(s[i]) : "v"(a));
    }
  }
  float r = 0;
  for (int i = 0; i < 8; i++) r += s[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount, grid = cus * 8, iters = 20000;
  float* d;
  hipMalloc(&d, grid * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, double wave_instr_per_iter) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD = grid*4 waves / (cus*4 SIMDs) = 8 ; instr per SIMD = 8 * iters * wave_instr_per_iter
    double per_simd = 8.0 * iters * wave_instr_per_iter;
    printf("%-28s %8.3f ms   %6.2f ns per wave-instruction per SIMD  (%.2f cycles @2.4GHz)\n", name, ms, ms * 1e6 / per_simd,
           ms * 1e6 / per_simd * 2.4);
  };
  printf("CUs %d clock %d kHz\n", cus, pr.clockRate);
  run("v_fma_f32 x16", [&] { hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f); }, 16);
  run("v_pk_fma_f32 x8", [&] { hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f); }, 8);
  run("v_mul_f32 x16", [&] { hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f); }, 16);
  run("v_pk_mul_f32 x8", [&] { hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f); }, 8);
  run("v_add_f32 x16", [&] { hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f); }, 16);
  run("v_pk_add_f32 x8", [&] { hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f); }, 8);
  run("v_fma_f64 x8", [&] { hipLaunchKernelGGL(k64, dim3(grid), dim3(256), 0, 0, (double*)d, iters, 1.0001, 0.5); }, 8);
  run("v_rcp_f32 x8", [&] { hipLaunchKernelGGL(ktrans<0>, dim3(grid), dim3(256), 0, 0, d, iters, 1.5f); }, 8);
  run("v_sqrt_f32 x8", [&] { hipLaunchKernelGGL(ktrans<1>, dim3(grid), dim3(256), 0, 0, d, iters, 1.5f); }, 8);
  run("v_cndmask_b32 x8", [&] { hipLaunchKernelGGL(ktrans<2>, dim3(grid), dim3(256), 0, 0, d, iters, 1.5f); }, 8);
  run("v_min_f32 x8", [&] { hipLaunchKernelGGL(ktrans<3>, dim3(grid), dim3(256), 0, 0, d, iters, 1.5f); }, 8);
  return 0;
}
