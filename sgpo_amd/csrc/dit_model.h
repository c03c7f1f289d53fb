// DiT model handle, shared by the forward (dit.hip) and the NOS step (nos.hip).
#pragma once
#include <vector>

#include "../../include/sgpo_hip.h"
#include "common.h"

struct DitBlockW {
  const float* norm1;
  const bf16_t* qkv;       // [3H][H]
  const bf16_t* attn_out;  // [H][H]
  const float* norm2;
  const bf16_t* fc1;       // [F][H]
  const float* fc1_b;
  const bf16_t* fc2;       // [H][F]
  const float* fc2_b;
};

struct sgpo_dit {
  sgpo_dit_config cfg;
  float* f32 = nullptr;     // all fp32 parameters (copy of the blob)
  bf16_t* b16 = nullptr;    // bf16 copies of the GEMM weights
  float* ada_w = nullptr;   // concatenated adaLN weights [n_mod][cond]
  float* ada_b = nullptr;   // [n_mod]
  float* rope_cos = nullptr;
  float* rope_sin = nullptr;
  int n_mod = 0;
  const float* embed;
  const float *s_w1, *s_b1, *s_w2, *s_b2;
  std::vector<DitBlockW> blocks;
  const float* out_norm;
  const bf16_t* out_lin;        // [vocab][H] bf16
  const float* out_lin_f32;     // [vocab][H] fp32 (NOS step)
  const float* out_lin_b;
};
