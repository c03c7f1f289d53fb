// prints the SIMD id of each wave of a 512-thread workgroup (diagnostic: which waves share a SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void k(int* out) {
  extern __shared__ char smem[];
  const int wave = threadIdx.x >> 6;
  const int hw = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);    // HW_ID: simd_id bits [5:4]
  const int wv = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);    // wave_id bits [3:0]
  if ((threadIdx.x & 63) == 0) { out[blockIdx.x * 16 + wave * 2] = hw; out[blockIdx.x * 16 + wave * 2 + 1] = wv; }
  if (threadIdx.x == 9999) smem[0] = 1;
}
int main() {
  int* d; hipMalloc(&d, 64 * 16 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
  hipLaunchKernelGGL(k, dim3(64), dim3(512), 140000, 0, d);
  int h[64 * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 6; ++b) { printf("block %d: simd", b); for (int w = 0; w < 8; ++w) printf(" %d", h[b * 16 + w * 2]); printf("  wave_id"); for (int w = 0; w < 8; ++w) printf(" %d", h[b * 16 + w * 2 + 1]); printf("\n"); }
  return 0;
}
