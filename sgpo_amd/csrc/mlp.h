// Value-function handle shared by the guidance (guidance.hip) and DAPS (daps.hip) kernels.
#pragma once
#include "common.h"

#define MH 64  // hidden width (problem/protein.py: hidden_dim 64)

struct sgpo_mlp {
  int n_read = 0, alphabet = 0, hidden = 0, time_cond = 0;
  size_t n_params = 0;         // floats in `params`
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

// ---- several stacked sampler calls in one launch (sgpo_guided_step_groups_mdlm) ----
// A group is one draw: rows [seq_lo, seq_lo + n_seq) of the stacked batch, its own Philox key and (guidance kernels)
// its own value function.  The kernels take up to SGPO_MAX_GROUPS of them by value and blockIdx.y picks one; inside a
// group every index is the one the group's own launch would use, so the results are those of separate launches.
#define SGPO_MAX_GROUPS 16
struct RsGroup {
  int seq_lo, n_seq, guided, pad_;
  uint64_t seed, offset, row_offset;
};
struct RsGroups {
  int n;   // 0: plain launch, the scalar arguments apply
  RsGroup g[SGPO_MAX_GROUPS];
};
int launch_reverse_step_mdlm_groups(const RsGroups& gs, int max_seq, const float* logits, const int64_t* x,
                                    const float* log_ratio, const int* slot_of_pos, int n_slots, int L, int K,
                                    int mask_index, float move_chance_t, float move_chance_s, float guide_temp,
                                    int64_t* x_out, int* nan_flag, hipStream_t stream, const uint64_t* key_dev = nullptr,
                                    int key_g0 = 0);
