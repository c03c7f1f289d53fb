// bf16 MFMA GEMM  C[M,N] = A[M,K] * W[N,K]^T  with fused epilogues (gfx950).
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2
// v_mfma_f32_32x32x16_bf16 tiles, fp32 accumulators.  Operand tiles go
// HBM/L2 -> LDS by LDS-DMA (global_load_lds, 16 B/lane) into a lane-linear image;
// the 16-B chunk swizzle  pc = c ^ ((row >> 1) & 7)  is applied on the SOURCE
// address and again on the ds_read_b128 (both-sides involution), which makes
// every 16-lane ds_read_b128 group hit 16 distinct 16-B slots (conflict-free).
// Two LDS buffers; tile t+1 streams in while tile t is multiplied.
#pragma once
#include "common.h"

enum GemmEpi {
  EPI_BF16 = 0,        // C bf16 = acc (+bias)
  EPI_QKV_ROPE = 1,    // C bf16 [M][3*hidden]; rotate-half RoPE on the q,k column sections
  EPI_GATE_RES = 2,    // Xout f32 = Xin + gate[b][n] * (acc + bias[n])
  EPI_BIAS_GELU = 3,   // C bf16 = gelu_tanh(acc + bias)
  EPI_BIAS_F32 = 4,    // C f32 = acc + bias   (n < N only; ldc given)
  EPI_BIAS_GELU_ERF = 5,  // C bf16 = gelu_erf(acc + bias)
  EPI_RES = 6,         // Xout f32 = Xin + acc + bias
  EPI_ACC_F32 = 7,     // C f32 += acc (+bias)  (multi-tap convolution accumulation)
  EPI_LN_GELU = 8,     // C bf16 = gelu_erf(LayerNorm(acc + bias) * ln_w + ln_b); N = 512 = one tile row (gemm_p.hip only)
  EPI_GATE_RES_LN = 9, // EPI_GATE_RES, and ln_out bf16 = adaLN-modulated LayerNorm of the new residual row (N = 512, gemm_p.hip only)
};

struct GemmParams {
  const bf16_t* A;     // [M][lda]
  const bf16_t* W;     // [N][ldw]
  void* C;             // output (bf16 or f32), [M][ldc]
  const float* bias;   // [N] or null
  const float* xin;    // residual input (f32, [M][ldc]) for EPI_GATE_RES / EPI_RES
  const float* gate;   // gate[b*gate_stride + n]
  const float* rope_cos;  // [max_len][head_dim/2]
  const float* rope_sin;
  int M, N, K;
  int lda, ldw, ldc;
  int L;               // tokens per sequence (row -> batch index / position)
  int B;               // sequences (clamp for padded rows)
  int gate_stride;     // 0 = one modulation row broadcast over the batch
  int rope_cols;       // columns [0, rope_cols) get RoPE (= 2*hidden)
  int head_dim;        // 64
  // A row remap for dilated convolution taps: source row = row + row_shift when it stays
  // inside the same sequence, else the operand row is treated as zero.
  int row_shift;
  // EPI_LN_GELU: affine LayerNorm over the N outputs of a row, then exact GELU
  const float* ln_w;
  const float* ln_b;
  float ln_eps;
  // EPI_GATE_RES_LN: LayerNorm weight in ln_w; ln_out[m][n] = ((x_new - mean) rstd ln_w[n]) (1 + ln_scale[b][n]) + ln_shift[b][n],
  // the modulation rows indexed like the gate (b * gate_stride)
  const float* ln_shift;
  const float* ln_scale;
  bf16_t* ln_out;
};

// Epilogue arithmetic shared by both GEMM kernels (gemm.hip, gemm_p.hip).  Every fused multiply-add is
// explicit so the two kernels round identically: a batch sharded into pieces small enough for the
// 128x128 kernel must reproduce the tokens of the unsharded batch bit for bit.
// __fmul_rn / __fadd_rn are plain * and + in the ROCm headers and contract like any other expression,
// so the uncontracted forms are spelled with the pragma.
__device__ __forceinline__ float rn_mul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float rn_add(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float epi_gelu_tanh(float x) {
  // 0.5 x (1 + tanh(u)) = x * sigmoid(2u),  2u = 2 sqrt(2/pi) (x + 0.044715 x^3) = x (c0 + c1 x^2);
  // sigmoid(2u) = 1 / (1 + exp2(x (k0 + k1 x^2))) with k = -c log2(e): 4 VALU + v_exp_f32 + v_rcp_f32
  const float t = fmaf(-0.10294324f, rn_mul(x, x), -2.30220819f);
  const float e = __builtin_amdgcn_exp2f(rn_mul(x, t));
  return rn_mul(x, __builtin_amdgcn_rcpf(1.0f + e));
}
// two elements at a time: the multiplies and the fma map to v_pk_mul_f32 / v_pk_fma_f32 (same IEEE results as the
// scalar form above, half the VALU issue slots); exp2 and rcp stay one lane-instruction per element
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ f32x2_t epi_gelu_tanh2(f32x2_t x) {
#pragma clang fp contract(off)
  const f32x2_t k1 = {-0.10294324f, -0.10294324f}, k0 = {-2.30220819f, -2.30220819f};
  const f32x2_t xx = x * x;
  const f32x2_t t = __builtin_elementwise_fma(k1, xx, k0);
  const f32x2_t a = x * t;
  f32x2_t r;
  r.x = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a.x));
  r.y = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a.y));
  return x * r;
}
__device__ __forceinline__ float epi_gelu_erf(float x) {
  return rn_mul(rn_mul(0.5f, x), 1.0f + erff(rn_mul(x, 0.7071067811865476f)));
}
// exact-GELU for an output that is rounded to bf16 right away: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, five
// fmas + v_rcp_f32 + v_exp_f32, a handful of registers -- erff's two-branch expansion does not fit beside 128
// accumulators).  The negative side is formed as x * 0.5 * (1 - erf|.|) = x * 0.5 * poly * exp(-z^2) without the
// cancellation of 1 + erf, so the tail keeps its relative accuracy.
__device__ __forceinline__ float epi_gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float pe = pl * t * __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);   // 1 - erf(z)
  return x * (x < 0.f ? 0.5f * pe : fmaf(-0.5f, pe, 1.0f));
}
__device__ __forceinline__ void epi_rope(float a, float b, float cs, float sn, float& o0, float& o1) {
  o0 = fmaf(a, cs, -rn_mul(b, sn));   // x1 cos - x2 sin
  o1 = fmaf(a, sn, rn_mul(b, cs));    // x1 sin + x2 cos
}
// x + gate * (acc + bias), the product rounded before the add (two roundings, never contracted to an
// fma): the 256x256 kernel forms the product in the accumulator layout and adds x after a transpose
// through LDS, and both kernels must give the same bits.  (The reference rounds the product to bf16.)
__device__ __forceinline__ float epi_gate_mul(float acc, float bias, float gate) { return rn_mul(acc + bias, gate); }
__device__ __forceinline__ float epi_gate_res(float acc, float bias, float gate, float xin) {
  return rn_add(xin, epi_gate_mul(acc, bias, gate));
}

struct GemmSeg {
  int n_seg;       // K = n_seg * seg_k ; segment s reads A rows shifted by shift[s]
  int seg_k;
  int shift[8];
};

int launch_gemm_bf16(const GemmParams& p, int epi, hipStream_t stream);
int launch_gemm_bf16_seg(const GemmParams& p, int epi, int n_seg, int seg_k, const int* shifts, hipStream_t stream);
// persistent 256x128 kernel (gemm_p.hip); returns SGPO_ERR_UNSUPPORTED when the shape does not fit it
int launch_gemm_persistent(const GemmParams& p, int epi, const GemmSeg& sg, const bf16_t* zero16, hipStream_t stream);
// GEMM + bias + LayerNorm + GELU in one kernel (full-row tiles); SGPO_ERR_UNSUPPORTED when the shape does not fit
// (N != 512, few rows): the caller then runs the EPI_BIAS_F32 GEMM and the separate LayerNorm + GELU pass
int launch_gemm_ln_gelu(const GemmParams& p, int n_seg, int seg_k, const int* shifts, hipStream_t stream);
// out-projection / fc2 GEMM + gate + fp32 residual + the NEXT LayerNorm-modulate in one kernel (same contract)
int launch_gemm_gate_res_ln(const GemmParams& p, hipStream_t stream);
int gemm_get_zero16(const bf16_t** out);
int gemm_cu_count();
