// Diagnostic: fp32 read-modify-write of a 1 GiB array, once per element: float4 load+store against
// coalesced no-return global_atomic_add_f32 (one 256-B row per wave instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
__global__ void rmw_ldst(float* x, size_t n4, float y) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = gridDim.x * (size_t)blockDim.x;
  for (; i < n4; i += st) { float4 v = reinterpret_cast<float4*>(x)[i]; v.x += y; v.y += y; v.z += y; v.w += y; reinterpret_cast<float4*>(x)[i] = v; }
}
__global__ void rmw_atomic(float* x, size_t n, float y) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = gridDim.x * (size_t)blockDim.x;
  for (; i < n; i += st) { float* p = x + i; asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(p), "v"(y) : "memory"); }
}
int main() {
  const size_t n = (size_t)256 << 20;   // 1 GiB
  float* x; hipMalloc(&x, n * 4); hipMemset(x, 0, n * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int grid : {2048, 8192, 65536}) {
    for (int mode = 0; mode < 2; ++mode) {
      float best = 1e9;
      for (int r = 0; r < 4; ++r) {
        hipEventRecord(a);
        if (mode == 0) rmw_ldst<<<grid, 256>>>(x, n / 4, 1.0f); else rmw_atomic<<<grid, 256>>>(x, n, 1.0f);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = fminf(best, ms);
      }
      printf("grid %6d %s: %.3f ms  (%.2f TB/s read+write)\n", grid, mode ? "atomic" : "ld/st ", best, 2.0 * n * 4 / best / 1e9);
    }
  }
  std::vector<float> h(1024); hipMemcpy(h.data(), x + 12345, 4096, hipMemcpyDeviceToHost);
  printf("check %.1f %.1f (expect 24)\n", h[0], h[1023]);
  return 0;
}
