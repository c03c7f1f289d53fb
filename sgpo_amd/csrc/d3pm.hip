// D3PM reverse-step (K x K transition-matrix posterior) and forward-noising kernels.
//
// Reference: models/d3pm.py:230-268 (p_sample), :45-52 (q_sample); SURVEY.md Appendix B:
//   p0 = softmax_fp32(z[:S]) -> fp64;  A[j] = DeltaQ[j][x];
//   p[j] = sum_i (A[j] * (p0[i] * Qbar'[i][j])) * p0[i]   (p0 enters squared, defect D9)
//   p /= sum_j p;  x' = argmax_j p[j] / e[j],  e ~ Exp(1)   (torch.multinomial)
// The reference materialises [N][L][S][S] fp64 (4.3 GB per GPU at cfg3); here one position lives in
// one 32-lane half-wave, Qbar' sits in LDS as fp64, and nothing but logits/tokens touches HBM:
// algorithmic bytes per position = 4*ld_logits + 8 (x in) + 8 (x out).
// fp64 arithmetic is kept (676 MACs per position: free) so soft outputs match the reference to ~1e-15.
#include "../../include/sgpo_hip.h"
#include "common.h"
#include "prof.h"

// fp64 half-wave (32-lane) all-reduces: four DPP steps on the two words of the double (lane^1, lane^2, mirror inside 8,
// mirror inside 16) and one swizzle across the two rows of 16 -- no ds_bpermute (two per exchanged double before).
#define D3_DPP_I(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)
__device__ __forceinline__ double d3_dpp(double v, int step) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  switch (step) {
    case 0: return __hiloint2double(D3_DPP_I(hi, 0xB1), D3_DPP_I(lo, 0xB1));
    case 1: return __hiloint2double(D3_DPP_I(hi, 0x4E), D3_DPP_I(lo, 0x4E));
    case 2: return __hiloint2double(D3_DPP_I(hi, 0x141), D3_DPP_I(lo, 0x141));
    case 3: return __hiloint2double(D3_DPP_I(hi, 0x140), D3_DPP_I(lo, 0x140));
    default: return __hiloint2double(__builtin_amdgcn_ds_swizzle(hi, 0x401F), __builtin_amdgcn_ds_swizzle(lo, 0x401F));
  }
}
__device__ __forceinline__ double half_sum_d(double v) {
#pragma unroll
  for (int st = 0; st < 5; ++st) v += d3_dpp(v, st);
  return v;
}
// lowest lane of the half holding the maximum (torch.argmax on CPU; inactive lanes carry -1)
__device__ __forceinline__ int half_argmax_d(double v, int k) {
  double m = v;
#pragma unroll
  for (int st = 0; st < 5; ++st) m = fmax(m, d3_dpp(m, st));
  const unsigned long long bal = __builtin_amdgcn_ballot_w64(v == m);
  const unsigned hm = (unsigned)(bal >> (threadIdx.x & 32));
  return hm ? __ffs(hm) - 1 : 0;
}

__global__ __launch_bounds__(256) void d3pm_posterior_kernel(
    const float* __restrict__ logits, int ld, const int64_t* __restrict__ x, int rows, int S,
    const float* __restrict__ delta_q, const float* __restrict__ q_bar, int step_mode, int noise_mode,
    const void* __restrict__ noise, uint64_t seed, uint64_t offset, uint64_t row_offset,
    int64_t* __restrict__ x_out, double* __restrict__ probs_out) {
  __shared__ double qb[32 * 32];
  __shared__ float dq[32 * 32];
  __shared__ double p0s[8][32];   // per half-wave: the row's p0, read back as broadcasts (was two ds_bpermute per term)
  for (int i = threadIdx.x; i < S * S; i += 256) {
    qb[(i / S) * 32 + (i % S)] = (double)q_bar[i];
    dq[(i / S) * 32 + (i % S)] = delta_q[i];
  }
  __syncthreads();
  const int j = threadIdx.x & 31;
  const bool act = j < S;
  // a workgroup walks many rows: staging the two S x S matrices (1352 loads + conversions) once per 8 rows was most of
  // the kernel's time
  for (int row = (blockIdx.x * 256 + threadIdx.x) >> 5; row < rows; row += gridDim.x * 8) {
  int xt = (int)x[row];
  xt = xt < 0 ? 0 : (xt >= S ? S - 1 : xt);
  // fp32 softmax over the first S logits (F.softmax(model(x,t)[:,:,:S]), models/d3pm.py:37)
  const float z = act ? logits[(size_t)row * ld + j] : -INFINITY;
  const float zmax = half_max(z);
  const float ez = act ? expf(z - zmax) : 0.f;
  const float p0f = ez / half_sum(ez);
  const double p0 = (double)p0f;
  const double A = act ? (double)dq[j * 32 + xt] : 0.0;  // Delta_Q.T[x][j]
  double acc = 0.0;
  double* ph = p0s[threadIdx.x >> 5];
  ph[j] = p0;                                   // written and read by the same wave: LDS keeps program order
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int i = 0; i < S; ++i) {
    const double pi = ph[i];
    const double b = pi * qb[i * 32 + (act ? j : 0)];
    acc += (A * b) * pi;
  }
  if (!act) acc = 0.0;
  const double p = acc / half_sum_d(acc);
  const size_t lidx = (size_t)row * S + j;
  if (probs_out && act) probs_out[lidx] = p;
  if (step_mode == SGPO_STEP_SOFT) continue;
  double den = 1.0;
  if (act) {
    if (noise_mode == SGPO_NOISE_EXPONENTIAL) den = reinterpret_cast<const double*>(noise)[lidx];
    else if (noise_mode == SGPO_NOISE_PHILOX) {
      Philox4 r = philox4x32_10(seed, ((uint64_t)row + row_offset) * (uint64_t)S + j, offset);
      den = (double)(-logf(u01_open(r.x)));
    }
  }
  const int samp = half_argmax_d(act ? p / den : -1.0, j);
  if (j == 0) x_out[row] = samp;
  }
}

extern "C" int sgpo_reverse_step_d3pm(const float* logits, int ld_logits, const int64_t* x, int N, int L, int S,
                                      const float* delta_q, const float* q_bar, int step_mode, int noise_mode,
                                      const void* noise, uint64_t seed, uint64_t offset, uint64_t row_offset,
                                      int64_t* x_out, double* probs_out, void* stream) {
  SGPO_REQUIRE(logits && x && delta_q && q_bar, "reverse_step_d3pm: null argument");
  SGPO_REQUIRE(N > 0 && L > 0 && S > 0 && S <= 32 && ld_logits >= S, "reverse_step_d3pm: N=%d L=%d S=%d ld=%d", N, L, S, ld_logits);
  SGPO_REQUIRE(step_mode == SGPO_STEP_SOFT || step_mode == SGPO_STEP_HARD, "reverse_step_d3pm: step_mode must be SOFT or HARD");
  SGPO_REQUIRE(step_mode == SGPO_STEP_SOFT ? probs_out != nullptr : x_out != nullptr, "reverse_step_d3pm: output missing");
  SGPO_REQUIRE(noise_mode == SGPO_NOISE_PHILOX || noise_mode == SGPO_NOISE_NONE || noise_mode == SGPO_NOISE_EXPONENTIAL,
               "reverse_step_d3pm: noise_mode must be PHILOX, NONE or EXPONENTIAL (fp64 draws)");
  SGPO_REQUIRE(noise_mode != SGPO_NOISE_EXPONENTIAL || step_mode == SGPO_STEP_SOFT || noise, "reverse_step_d3pm: noise tensor missing");
  const int rows = N * L;
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof_scope(PK_D3PM_POSTERIOR, st, (double)rows * (ld_logits * 4.0 + 16.0));
  int blocks = (rows + 7) / 8;
  if (blocks > 4096) blocks = 4096;   // 16 workgroups per CU; each walks rows with a grid stride
  hipLaunchKernelGGL(d3pm_posterior_kernel, dim3(blocks), dim3(256), 0, st, logits, ld_logits, x, rows, S, delta_q,
                     q_bar, step_mode, noise_mode, noise, seed, offset, row_offset, x_out, probs_out);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

// q_sample: prob[j] = Qbar[t_b][j][x]  (bmm(Qbar[t], onehot(x)^T), models/d3pm.py:49-51) -> multinomial, fp32
__global__ __launch_bounds__(256) void d3pm_q_sample_kernel(const int64_t* __restrict__ x, const int64_t* __restrict__ t,
                                                            const float* __restrict__ q_bar_all, int T, int rows, int L,
                                                            int S, int noise_mode, const float* __restrict__ noise,
                                                            uint64_t seed, uint64_t offset, uint64_t row_offset,
                                                            int64_t* __restrict__ x_out) {
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 5;
  const int j = threadIdx.x & 31;
  if (row >= rows) return;
  const bool act = j < S;
  int xt = (int)x[row];
  xt = xt < 0 ? 0 : (xt >= S ? S - 1 : xt);
  int64_t tb = t[row / L];
  tb = tb < 0 ? 0 : (tb >= T ? T - 1 : tb);
  const float p = act ? q_bar_all[((size_t)tb * S + j) * S + xt] : 0.f;
  float den = 1.f;
  const size_t lidx = (size_t)row * S + j;
  if (act) {
    if (noise_mode == SGPO_NOISE_EXPONENTIAL) den = noise[lidx];
    else if (noise_mode == SGPO_NOISE_PHILOX) {
      Philox4 r = philox4x32_10(seed, ((uint64_t)row + row_offset) * (uint64_t)S + j, offset);
      den = -logf(u01_open(r.x));
    }
  }
  float v = act ? p / den : -INFINITY;
  int k = j;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    float ov = __shfl_xor(v, o);
    int ok = __shfl_xor(k, o);
    if (ov > v || (ov == v && ok < k)) { v = ov; k = ok; }
  }
  if (j == 0) x_out[row] = k;
}

extern "C" int sgpo_q_sample_d3pm(const int64_t* x, const int64_t* t, const float* q_bar_all, int T, int B, int L, int S,
                                  int noise_mode, const float* noise, uint64_t seed, uint64_t offset,
                                  uint64_t row_offset, int64_t* x_out, void* stream) {
  SGPO_REQUIRE(x && t && q_bar_all && x_out, "q_sample_d3pm: null argument");
  SGPO_REQUIRE(B > 0 && L > 0 && S > 0 && S <= 32 && T > 0, "q_sample_d3pm: B=%d L=%d S=%d T=%d", B, L, S, T);
  SGPO_REQUIRE(noise_mode == SGPO_NOISE_PHILOX || (noise_mode == SGPO_NOISE_EXPONENTIAL && noise), "q_sample_d3pm: noise_mode must be PHILOX or EXPONENTIAL (+tensor)");
  const int rows = B * L;
  hipLaunchKernelGGL(d3pm_q_sample_kernel, dim3((rows + 7) / 8), dim3(256), 0, (hipStream_t)stream, x, t, q_bar_all, T,
                     rows, L, S, noise_mode, noise, seed, offset, row_offset, x_out);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}
