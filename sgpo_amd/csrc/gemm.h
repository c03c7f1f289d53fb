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
};

int launch_gemm_bf16(const GemmParams& p, int epi, hipStream_t stream);
