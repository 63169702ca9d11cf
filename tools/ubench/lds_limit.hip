// How much dynamic LDS can one workgroup get on gfx950, with and without hipFuncSetAttribute, through the runtime
// launch and through hipModuleLaunchKernel? (prune_top_kernel wants ~120 KB for one 1024-thread workgroup.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ float smem[];
extern "C" __global__ void touch(float* out, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) smem[i] = (float)i;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = smem[n - 1];
}
int main() {
  float* d;
  hipMalloc(&d, 4);
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu\n", pr.sharedMemPerBlock, pr.maxSharedMemoryPerMultiProcessor);
  for (int kb : {48, 64, 65, 96, 128, 160}) {
    const size_t bytes = (size_t)kb * 1024;
    hipLaunchKernelGGL(touch, dim3(1), dim3(1024), bytes, 0, d, (int)(bytes / 4));
    hipError_t e1 = hipGetLastError();
    hipError_t e2 = hipDeviceSynchronize();
    float h = 0;
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("plain  %3d KB: launch %s sync %s value %.0f\n", kb, hipGetErrorName(e1), hipGetErrorName(e2), h);
  }
  hipError_t ea = hipFuncSetAttribute((const void*)touch, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  printf("hipFuncSetAttribute -> %s\n", hipGetErrorName(ea));
  for (int kb : {96, 128, 160}) {
    const size_t bytes = (size_t)kb * 1024;
    hipLaunchKernelGGL(touch, dim3(1), dim3(1024), bytes, 0, d, (int)(bytes / 4));
    hipError_t e1 = hipGetLastError();
    hipError_t e2 = hipDeviceSynchronize();
    float h = 0;
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("attr   %3d KB: launch %s sync %s value %.0f\n", kb, hipGetErrorName(e1), hipGetErrorName(e2), h);
  }
  return 0;
}
