// Value-function handle shared by the guidance (guidance.hip) and DAPS (daps.hip) kernels.
#pragma once
#include "common.h"

#define MH 64  // hidden width (problem/protein.py: hidden_dim 64)

struct sgpo_mlp {
  int n_read = 0, alphabet = 0, hidden = 0, time_cond = 0;
  int* read_pos = nullptr;     // [n_read] device
  int* read_pos_host = nullptr;
  float* params = nullptr;     // raw blob copy
  float* T1 = nullptr;         // [n_read][alphabet][64]
  float* be1 = nullptr;        // [64] b1 + sum_i W1_i be
  float* w1t_time = nullptr;   // [k][j] = W1[j][n_read*64 + k]
  float* wt_lin_t = nullptr;   // [k][j] = time_embedder.1.weight[j][k]
  float* w_t[3] = {nullptr, nullptr, nullptr};  // transposed 64x64 layers (mlp.2, mlp.4, cls_head.0): [k][j]
  float* w_perm = nullptr;     // the three layers permuted for the MFMA A operand: [layer][to][ti][r][h][i]
  float* bias_perm = nullptr;  // [layer][t][r][h] biases in accumulator order (3 layers) + w5 in the same order
  const float *W2, *W3, *W4;   // row-major [out][in] views into params
  const float *tW, *bt, *b2, *b3, *b4, *w5, *b5;
};
