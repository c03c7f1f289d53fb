// DAPS discrete Metropolis-Hastings inner loop: `steps` sequential accept/reject moves per
// sequence in ONE launch (reference: sampling/daps.py:68-114, 1000 PyTorch-dispatched steps).
//
// One wave per sequence, lane j = hidden unit j of the MLP value function (problem/protein.py:
// 58-121, evaluated at t = 0 via log_likelihood, :37-38).  The three 64x64 layers keep their
// weight rows in VGPRs (192 registers); activations cross lanes by v_readlane (scalar operand of
// the FMA), so a step touches no LDS and no HBM except the L2-resident first-layer table
// T1[i][token][:] (see guidance.hip).  Every per-chain quantity (likelihood, Hamming distance,
// accept decision) is wave-uniform.  Latency-bound by construction: report steps/s, not a roofline.
//
// Bug-compatible details kept: proposals index the 0-indexed `residues`, the value function reads
// the raw (1-indexed) config list (defect D3, both lists are data here); `seed == rho` drops both
// branches so the whole row becomes token 0 with likelihood 0 and distance 0 (daps.py:110-112).
#include "../../include/sgpo_hip.h"
#include "common.h"
#include "mlp.h"
#include "prof.h"

#define MAX_READ 512
#define MAX_RES 512

__device__ __forceinline__ float bcast(float v, int k) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k));
}

__device__ __forceinline__ float dense64(const float (&w)[64], float bias, float a) {
  float acc = bias;
#pragma unroll
  for (int k = 0; k < 64; ++k) acc = fmaf(w[k], bcast(a, k), acc);
  return acc;
}

// c0[j] = be1[j] + W1_time[j][:] . time_embedder(t)   (constant part of the first-layer pre-activation)
__global__ __launch_bounds__(64) void mlp_const_kernel(const float* __restrict__ be1, const float* __restrict__ tW,
                                                       const float* __restrict__ wt_lin_t, const float* __restrict__ bt,
                                                       const float* __restrict__ w1t_time, int time_cond, float t,
                                                       float* __restrict__ c0) {
  __shared__ float buf[2][64];
  const int j = threadIdx.x;
  float pre = be1[j];
  if (time_cond) {
    const float arg = t * tW[j & 31] * 2.0f * 3.14159274101257324f;
    buf[0][j] = (j < 32) ? sinf(arg) : cosf(arg);
    __syncthreads();
    float te = bt[j];
    for (int k = 0; k < 64; ++k) te = fmaf(wt_lin_t[k * 64 + j], buf[0][k], te);
    buf[1][j] = te;
    __syncthreads();
    for (int k = 0; k < 64; ++k) pre = fmaf(w1t_time[k * 64 + j], buf[1][k], pre);
  }
  c0[j] = pre;
}

struct MhArgs {
  const int64_t* x0hat; int64_t* x_out;
  const float* T1; const int* read_pos; const float* c0;
  const float *W2, *W3, *W4, *b2, *b3, *b4, *w5, *b5;
  const int* residues;   // [n_res] 0-indexed mutable positions
  const int* res_slot;   // [n_res] read slot of that position or -1
  const int *r_idx, *r_val; const float* r_u;  // injected noise [steps][N] or null
  int N, L, S, alphabet, n_read, n_res, steps, noise_mode;
  float alpha, log_alpha, log_1alpha;
  uint64_t seed, offset, row_offset;
};

__global__ __launch_bounds__(256, 2) void daps_mh_kernel(MhArgs a) {
  __shared__ int s_tok[4][MAX_READ];    // tokens at the read slots (current state)
  __shared__ int s_res[4][MAX_RES];     // tokens at the mutable residues (current state)
  __shared__ int s_ref[4][MAX_RES];     // x0hat at the mutable residues
  const int wave = threadIdx.x >> 6, j = threadIdx.x & 63;
  const int seq = blockIdx.x * 4 + wave;
  if (seq >= a.N) return;
  int* tok = s_tok[wave];
  int* res = s_res[wave];
  int* ref = s_ref[wave];
  const int64_t* xr = a.x0hat + (size_t)seq * a.L;
  for (int i = j; i < a.n_read; i += 64) {
    int t = (int)xr[a.read_pos[i]];
    tok[i] = t < 0 ? 0 : (t >= a.alphabet ? a.alphabet - 1 : t);
  }
  for (int r = j; r < a.n_res; r += 64) {
    int t = (int)xr[a.residues[r]];
    res[r] = t; ref[r] = t;
  }
  float w2[64], w3[64], w4[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) {
    w2[k] = a.W2[j * 64 + k];
    w3[k] = a.W3[j * 64 + k];
    w4[k] = a.W4[j * 64 + k];
  }
  const float b2 = a.b2[j], b3 = a.b3[j], b4 = a.b4[j], w5 = a.w5[j], b5 = a.b5[0], c0 = a.c0[j];

  // f(state with slot `sl` overridden by token `v`; sl < 0: current state)
  auto loglik = [&](int sl, int v) -> float {
    float pre = c0;
    for (int i = 0; i < a.n_read; ++i) {
      const int t = (i == sl) ? v : tok[i];
      pre += a.T1[((size_t)i * a.alphabet + t) * 64 + j];
    }
    float h = fmaxf(pre, 0.f);
    h = fmaxf(dense64(w2, b2, h), 0.f);
    h = dense64(w3, b3, h);
    h = fmaxf(dense64(w4, b4, h), 0.f);
    return wave_sum(h * w5) + b5;
  };

  float cur_ll = loglik(-1, 0) * a.alpha;
  int cur_hm = 0;
  bool zeroed = false;
  int nz_nonres = 0;  // #{non-residue positions with x0hat != 0}; only meaningful once zeroed

  for (int s = 0; s < a.steps; ++s) {
    int r, v;
    float u;
    if (a.noise_mode == SGPO_NOISE_PHILOX) {
      Philox4 p = philox4x32_10(a.seed, (uint64_t)seq + a.row_offset, (a.offset << 20) | (uint64_t)s);
      r = (int)(p.x % (uint32_t)a.n_res);
      v = (int)(p.y % (uint32_t)a.S);
      u = (float)(p.z >> 8) * (1.0f / 16777216.0f);
    } else {
      r = a.r_idx[(size_t)s * a.N + seq];
      r = r < 0 ? 0 : (r >= a.n_res ? a.n_res - 1 : r);   // injected indices address LDS and global tables
      v = a.r_val[(size_t)s * a.N + seq];
      u = a.r_u[(size_t)s * a.N + seq];
    }
    const int sl = a.res_slot[r];
    const int old = res[r];
    int vt = v < 0 ? 0 : (v >= a.alphabet ? a.alphabet - 1 : v);
    const float ll = loglik(sl, vt) * a.alpha;
    int hm = cur_hm - (old != ref[r] ? 1 : 0) + (v != ref[r] ? 1 : 0);
    if (zeroed) {
      // the reference restarts current_hm at 0 after the double-drop but measures proposals on the full row
      int part = 0;
      for (int q = j; q < a.n_res; q += 64) part += ((q == r ? v : res[q]) != ref[q]) ? 1 : 0;
      hm = nz_nonres + (int)wave_sum((float)part);
    }
    float lr = ll - cur_ll;
    lr += ((float)hm * a.log_alpha + (float)(a.L - hm) * a.log_1alpha) -
          ((float)cur_hm * a.log_alpha + (float)(a.L - cur_hm) * a.log_1alpha);
    float rho = expf(lr);
    if (rho == rho) rho = fminf(rho, 1.0f);
    if (u < rho) {           // accept
      __builtin_amdgcn_wave_barrier();
      if (j == 0) { res[r] = v; if (sl >= 0) tok[sl] = vt; }
      __builtin_amdgcn_wave_barrier();
      cur_ll = ll; cur_hm = hm;
    } else if (!(u > rho)) {  // seed == rho (or NaN): both masks false -> row, likelihood, distance all 0
      if (!zeroed) {
        // count non-residue positions where x0hat != 0
        int cnt = 0;
        for (int l = j; l < a.L; l += 64) cnt += (xr[l] != 0) ? 1 : 0;
        int rescnt = 0;
        for (int q = j; q < a.n_res; q += 64) rescnt += (ref[q] != 0) ? 1 : 0;
        nz_nonres = (int)wave_sum((float)cnt) - (int)wave_sum((float)rescnt);
        zeroed = true;
      }
      __builtin_amdgcn_wave_barrier();
      for (int q = j; q < a.n_res; q += 64) res[q] = 0;
      for (int i = j; i < a.n_read; i += 64) tok[i] = 0;
      __builtin_amdgcn_wave_barrier();
      cur_ll = 0.f; cur_hm = 0;
    }
  }
  // ---- write the row: x0hat (or zeros) with the mutable residues replaced
  int64_t* xo = a.x_out + (size_t)seq * a.L;
  for (int l = j; l < a.L; l += 64) xo[l] = zeroed ? 0 : xr[l];
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  for (int q = j; q < a.n_res; q += 64) xo[a.residues[q]] = res[q];
}

extern "C" size_t sgpo_daps_mh_workspace_bytes(void) { return 256; }

// residues / res_slot: DEVICE int32 arrays [n_res] (mutable positions, 0-indexed, and the value function's
// read slot of each or -1); log_alpha / log_1alpha: the two scalars of DAPS.log_ratio (daps.py:69-71),
// computed by the host in fp32 exactly as torch does.
extern "C" int sgpo_daps_mh(sgpo_mlp* m, const int64_t* x0hat, int N, int L, int S, const int* residues,
                            const int* res_slot, int n_res, float log_alpha, float log_1alpha, float alpha, int steps,
                            int noise_mode, const int* r_idx, const int* r_val, const float* r_u, uint64_t seed,
                            uint64_t offset, uint64_t row_offset, int64_t* x_out, void* workspace,
                            size_t workspace_bytes, void* stream) {
  SGPO_REQUIRE(m && x0hat && residues && res_slot && x_out && workspace, "daps_mh: null argument");
  SGPO_REQUIRE(N > 0 && L > 0 && S > 0 && S <= m->alphabet && steps >= 0, "daps_mh: N=%d L=%d S=%d steps=%d", N, L, S, steps);
  SGPO_REQUIRE(n_res > 0 && n_res <= MAX_RES && m->n_read <= MAX_READ, "daps_mh: n_res=%d (max %d), n_read=%d (max %d)", n_res, MAX_RES, m->n_read, MAX_READ);
  SGPO_REQUIRE(steps < (1 << 20), "daps_mh: steps must be < 2^20");
  SGPO_REQUIRE(noise_mode == SGPO_NOISE_PHILOX || (r_idx && r_val && r_u), "daps_mh: injected noise arrays missing");
  SGPO_REQUIRE(x_out != x0hat, "daps_mh: x_out must not alias x0hat");
  if (workspace_bytes < sgpo_daps_mh_workspace_bytes()) {
    sgpo_set_error("daps_mh: workspace %zu < required %zu bytes", workspace_bytes, sgpo_daps_mh_workspace_bytes());
    return SGPO_ERR_WORKSPACE;
  }
  for (int i = 0; i < m->n_read; ++i)
    SGPO_REQUIRE(m->read_pos_host[i] < L, "daps_mh: value function reads position %d of a length-%d row", m->read_pos_host[i], L);
  hipStream_t st = (hipStream_t)stream;
  float* c0 = (float*)workspace;
  hipLaunchKernelGGL(mlp_const_kernel, dim3(1), dim3(64), 0, st, m->be1, m->tW, m->wt_lin_t, m->bt, m->w1t_time,
                     m->time_cond, 0.0f, c0);
  MhArgs a;
  a.x0hat = x0hat; a.x_out = x_out; a.T1 = m->T1; a.read_pos = m->read_pos; a.c0 = c0;
  a.W2 = m->W2; a.W3 = m->W3; a.W4 = m->W4; a.b2 = m->b2; a.b3 = m->b3; a.b4 = m->b4; a.w5 = m->w5; a.b5 = m->b5;
  a.residues = residues; a.res_slot = res_slot; a.r_idx = r_idx; a.r_val = r_val; a.r_u = r_u;
  a.N = N; a.L = L; a.S = S; a.alphabet = m->alphabet; a.n_read = m->n_read; a.n_res = n_res; a.steps = steps;
  a.noise_mode = noise_mode; a.alpha = alpha; a.log_alpha = log_alpha; a.log_1alpha = log_1alpha;
  a.seed = seed; a.offset = offset; a.row_offset = row_offset;
  ProfScope prof_scope(PK_DAPS_MH, st, (double)N * steps);
  hipLaunchKernelGGL(daps_mh_kernel, dim3((N + 3) / 4), dim3(256), 0, st, a);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}
