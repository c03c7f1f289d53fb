// Diagnostic: one fused-epilogue GEMM at large M (256x256 tiles) against the same rows computed in small calls.
#include <hip/hip_runtime.h>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include "../../sgpo_amd/csrc/gemm.h"
static unsigned short f2b(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
int main(int argc, char** argv) {
  const int epi = argc > 1 ? atoi(argv[1]) : EPI_GATE_RES;
  const int M = 4284, N = 512, K = argc > 2 ? atoi(argv[2]) : 512, L = 119, B = M / L;
  std::vector<unsigned short> A((size_t)M * K), W((size_t)N * K);
  std::vector<float> bias(N), gate((size_t)B * N), x((size_t)M * N);
  srand(1);
  auto rnd = [] { return (rand() / (float)RAND_MAX) * 2.f - 1.f; };
  for (auto& v : A) v = f2b(rnd());
  for (auto& v : W) v = f2b(rnd() * 0.05f);
  for (auto& v : bias) v = rnd() * 0.1f;
  for (auto& v : gate) v = rnd();
  for (auto& v : x) v = rnd();
  unsigned short *dA, *dW; float *db, *dg, *dx1, *dx2;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dW, W.size() * 2); hipMalloc(&db, N * 4); hipMalloc(&dg, gate.size() * 4);
  hipMalloc(&dx1, x.size() * 4); hipMalloc(&dx2, x.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), W.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(db, bias.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(dg, gate.data(), gate.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx1, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dx2, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  GemmParams p{};
  p.A = dA; p.W = dW; p.bias = db; p.gate = dg; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldw = K; p.ldc = N; p.L = L; p.B = B;
  p.gate_stride = N; p.head_dim = 64;
  p.C = dx1; p.xin = dx1;
  int rc = launch_gemm_bf16(p, epi, 0);
  printf("big rc=%d\n", rc);
  const int chunk = 4 * L;   // 476 rows -> 128x128 kernel
  for (int m0 = 0; m0 < M; m0 += chunk) {
    GemmParams q = p;
    const int mm = (M - m0 < chunk) ? M - m0 : chunk;
    q.A = dA + (size_t)m0 * K; q.C = dx2 + (size_t)m0 * N; q.xin = dx2 + (size_t)m0 * N; q.M = mm; q.B = mm / L;
    q.gate = dg + (size_t)(m0 / L) * N;
    rc = launch_gemm_bf16(q, epi, 0);
    if (rc) printf("small rc=%d\n", rc);
  }
  hipDeviceSynchronize();
  std::vector<float> o1(x.size()), o2(x.size());
  hipMemcpy(o1.data(), dx1, x.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), dx2, x.size() * 4, hipMemcpyDeviceToHost);
  size_t nd = 0; double mx = 0; size_t first = 0;
  for (size_t i = 0; i < o1.size(); ++i) if (o1[i] != o2[i]) { if (!nd) first = i; ++nd; mx = fmax(mx, fabs(o1[i] - o2[i])); }
  printf("epi %d K %d: %zu of %zu differ, max %.3g, first at row %zu col %zu (%.8g vs %.8g)\n", epi, K, nd, o1.size(), mx, first / N, first % N,
         nd ? o1[first] : 0.f, nd ? o2[first] : 0.f);
  return 0;
}
