// dumps HW_ID / LDS_ALLOC / XCC_ID per workgroup (experiment support; not part of the product)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(256) k(uint32_t* out) {
  extern __shared__ char smem[];
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11));
    out[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
    out[blockIdx.x * 4 + 3] = (uint32_t)(uintptr_t)smem;
  }
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(100);
}
int main() {
  uint32_t* d; const int n = 1024;
  hipMalloc(&d, n * 16);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 65536, 0, d);
  uint32_t* h = (uint32_t*)malloc(n * 16);
  hipMemcpy(h, d, n * 16, hipMemcpyDeviceToHost);
  for (int i = 0; i < 40; ++i) printf("wg %d hw_id %08x lds_alloc %08x xcc %08x smem %x\n", i, h[i*4], h[i*4+1], h[i*4+2], h[i*4+3]);
  for (int i = 512; i < 520; ++i) printf("wg %d hw_id %08x lds_alloc %08x xcc %08x smem %x\n", i, h[i*4], h[i*4+1], h[i*4+2], h[i*4+3]);
  return 0;
}
