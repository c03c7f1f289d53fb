// Diagnostic: the attention kernel alone at the bench shape, optionally with parts removed (-DATT_ABL=1|2|4).
// Build (standalone: linking libsgpo_hip.so as well would register the same kernel twice and run the library's copy):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DATT_ABL=n tools/dbg/att_bench.hip sgpo_amd/csrc/prof.hip -o tools/dbg/att_bench_n
#include "../../sgpo_amd/csrc/dit_kernels.hip"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
extern "C" void sgpo_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096, L = argc > 2 ? atoi(argv[2]) : 119, H = 8, reps = 20;
  const size_t n_in = (size_t)B * L * 3 * H * 64, n_out = (size_t)B * L * H * 64;
  std::vector<unsigned short> h(n_in);
  unsigned lcg = 1;
  for (auto& v : h) { lcg = lcg * 1664525u + 1013904223u; v = (unsigned short)((0x3c00 + ((lcg >> 8) & 0x3ff)) ^ ((lcg >> 31) << 15)); }   // |x| in [0.0078, 0.0156) bf16
  bf16_t *qkv, *out;
  if (hipMalloc(&qkv, n_in * 2) != hipSuccess || hipMalloc(&out, n_out * 2) != hipSuccess) return 1;
  hipMemcpy(qkv, h.data(), n_in * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) if (launch_attention(qkv, out, B, L, H, 64, 0)) return 2;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) launch_attention(qkv, out, B, L, H, 64, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("ATT_ABL=%d B=%d L=%d: %.1f us per launch, %.2f TB/s algorithmic\n", ATT_ABL, B, L, 1e3 * ms / reps,
         (n_in + n_out) * 2.0 / (1e-3 * ms / reps) / 1e12);
  return 0;
}
