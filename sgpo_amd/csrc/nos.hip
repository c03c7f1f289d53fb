// NOS (hidden-state guidance) step for MDLM: MDLM._nos_denoise (models/mdlm.py:184-338) with the
// DITClassifier head on x_emb (dit.py:494-515: mean over L, Linear(hidden, 1)).
//
// The reference runs torch autograd through final-layer + KLDiv + Adagrad `num_nos_steps` times.
// Only the final layer depends on delta and every row (b, l) is independent, so the backward is
// closed form and one wave does a whole inner step for one row, in fp32:
//   x = h + delta;  y = LN(x);  xn = y * g + shift   (g = norm_w * (1 + scale))
//   z = W xn + b;   p = softmax(z without the mask column)
//   dL/dz[j] = coef * (1 - r) / B * (p[j] - p0[j])   for masked rows, j != mask;  0 otherwise
//       (r = move_chance_s / move_chance_t; p0 = the same softmax at delta = 0; KLDivLoss(batchmean))
//   dxn = W^T dz;  dy = dxn * g;  dx = rstd * (dy - mean(dy) - y * mean(dy * y))
//   grad = dx - w_cls / L            (d(-target)/d delta: the mean-pool + Linear head)
//   Adagrad: G += grad^2;  delta -= lr * grad / (sqrt(G) + 1e-10)
// Final: logits(h + delta) -> fused reverse-step kernel (HARD, move chances (1, r)) which gives
// guided_probs = exp(log p + log(1 - r)), mask column r, _sample_categorical and the carry-over.
// HBM per row and step: h, delta, G read + delta, G written = 20 H bytes; W (vocab x H fp32) in LDS.
#include "../../include/sgpo_hip.h"
#include "common.h"
#include "dit_model.h"
#include "prof.h"

const float* dit_ws_final_mod(const sgpo_dit* m, int B, int L, void* workspace);

enum { NOS_INIT = 0, NOS_STEP = 1, NOS_FINAL = 2 };

struct NosArgs {
  const float* h; float* delta; float* G; float* p0; float* logits_out;
  const int64_t* xt;
  const float *W, *bias, *norm_w, *shift, *scale, *w_cls;
  int mod_stride, rows, L, B, V, mask_index;
  float cscale, lr, inv_L;
};

template <int H, int MODE>
__global__ __launch_bounds__(512) void nos_row_kernel(NosArgs a) {   // 8 waves share one copy of W in LDS
  extern __shared__ float wl[];  // [V][H]
  constexpr int PER = H / 64;
  for (int i = threadIdx.x; i < a.V * H; i += 512) wl[i] = a.W[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 8 + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * 8;
  for (int row = wid; row < a.rows; row += n_waves) {
    int b = row / a.L;
    if (b > a.B - 1) b = a.B - 1;
    const float* sh = a.shift + (size_t)b * a.mod_stride;
    const float* sc = a.scale + (size_t)b * a.mod_stride;
    // Unmasked rows (wave-uniform): dL/dz is 0 there, so neither p0 nor the logits of the inner steps are ever
    // used and the gradient is the constant -w_cls / L: only the Adagrad update remains (16 H bytes, no LDS
    // traffic).  Same bits as the full computation (0 - c == -c).  The final logits are formed for every row.
    const bool masked = (int)a.xt[row] == a.mask_index;
    if (MODE != NOS_FINAL && !masked) {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int col = i * 64 + lane;
        const size_t idx = (size_t)row * H + col;
        if (MODE == NOS_INIT) {
          a.delta[idx] = 0.f;
          a.G[idx] = 0.f;
        } else {
          const float grad = 0.f - a.w_cls[col] * a.inv_L;
          const float Gn = a.G[idx] + grad * grad;
          a.G[idx] = Gn;
          a.delta[idx] = a.delta[idx] - a.lr * grad / (sqrtf(Gn) + 1e-10f);
        }
      }
      continue;
    }
    float x[PER], y[PER], g[PER];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int col = i * 64 + lane;
      x[i] = a.h[(size_t)row * H + col];
      if (MODE != NOS_INIT) x[i] += a.delta[(size_t)row * H + col];
      s += x[i];
    }
    const float mean = wave_sum(s) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const float d = x[i] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / H) + 1e-5f);
    float xn[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int col = i * 64 + lane;
      y[i] = (x[i] - mean) * rstd;
      g[i] = a.norm_w[col] * (1.f + sc[col]);
      xn[i] = y[i] * g[i] + sh[col];
    }
    // logits: per-lane partial dot products, then a reduce-scatter over the wave (32 shuffles instead of 6 per
    // logit): lane l ends up with the total of logit k = (l >> 1) & 31 (both lanes of a pair hold it)
    float acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      acc[k] = 0.f;
      if (k < a.V) {
#pragma unroll
        for (int i = 0; i < PER; ++i) acc[k] = fmaf(wl[k * H + i * 64 + lane], xn[i], acc[k]);
      }
    }
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
    float r16[16], r8[8], r4[4], r2[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) r16[i] = (b5 ? acc[16 + i] : acc[i]) + __shfl_xor(b5 ? acc[i] : acc[16 + i], 32);
#pragma unroll
    for (int i = 0; i < 8; ++i) r8[i] = (b4 ? r16[8 + i] : r16[i]) + __shfl_xor(b4 ? r16[i] : r16[8 + i], 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) r4[i] = (b3 ? r8[4 + i] : r8[i]) + __shfl_xor(b3 ? r8[i] : r8[4 + i], 8);
#pragma unroll
    for (int i = 0; i < 2; ++i) r2[i] = (b2 ? r4[2 + i] : r4[i]) + __shfl_xor(b2 ? r4[i] : r4[2 + i], 4);
    float zl = (b1 ? r2[1] : r2[0]) + __shfl_xor(b1 ? r2[0] : r2[1], 2);
    zl += __shfl_xor(zl, 1);
    const int kl = (lane >> 1) & 31;
    const bool first = (lane & 1) == 0;
    zl = kl < a.V ? zl + a.bias[kl] : -INFINITY;
    if (MODE == NOS_FINAL) {
      if (first && kl < a.V) a.logits_out[(size_t)row * a.V + kl] = zl;
      continue;
    }
    // softmax over the non-mask columns (subs parameterisation: mask logit += -1e6 -> weight 0)
    const bool valid = kl < a.V && kl != a.mask_index;
    const float zmax = wave_max(valid ? zl : -INFINITY);
    const float e = valid ? expf(zl - zmax) : 0.f;
    const float inv = 1.0f / wave_sum(first ? e : 0.f);
    const float pl = e * inv;
    if (MODE == NOS_INIT) {
      if (first) a.p0[(size_t)row * 32 + kl] = pl;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        a.delta[(size_t)row * H + i * 64 + lane] = 0.f;
        a.G[(size_t)row * H + i * 64 + lane] = 0.f;
      }
      continue;
    }
    // ---- NOS_STEP: closed-form backward + Adagrad (masked rows only get here)
    float dx[PER];
    {
      float dxn[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) dxn[i] = 0.f;
      const float dzl = valid ? a.cscale * (pl - a.p0[(size_t)row * 32 + kl]) : 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        if (k < a.V && k != a.mask_index) {
          const float dz = __shfl(dzl, 2 * k);
#pragma unroll
          for (int i = 0; i < PER; ++i) dxn[i] = fmaf(wl[k * H + i * 64 + lane], dz, dxn[i]);
        }
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < PER; ++i) { dxn[i] *= g[i]; s1 += dxn[i]; s2 += dxn[i] * y[i]; }
      const float m1 = wave_sum(s1) * (1.0f / H), m2 = wave_sum(s2) * (1.0f / H);
#pragma unroll
      for (int i = 0; i < PER; ++i) dx[i] = rstd * (dxn[i] - m1 - y[i] * m2);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int col = i * 64 + lane;
      const size_t idx = (size_t)row * H + col;
      const float grad = dx[i] - a.w_cls[col] * a.inv_L;
      const float Gn = a.G[idx] + grad * grad;
      a.G[idx] = Gn;
      a.delta[idx] = a.delta[idx] - a.lr * grad / (sqrtf(Gn) + 1e-10f);
    }
  }
}

template <int H>
static int launch_nos(int mode, const NosArgs& a, hipStream_t st) {
  const size_t lds = (size_t)a.V * H * sizeof(float);
  int blocks = (a.rows + 7) / 8;
  if (blocks > 512) blocks = 512;
#define L_(M)                                                                                         \
  case M: {                                                                                           \
    static bool set = false;                                                                          \
    if (!set) {                                                                                       \
      SGPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nos_row_kernel<H, M>),        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
      set = true;                                                                                     \
    }                                                                                                 \
    hipLaunchKernelGGL((nos_row_kernel<H, M>), dim3(blocks), dim3(512), lds, st, a);                  \
  } break;
  switch (mode) { L_(NOS_INIT) L_(NOS_STEP) L_(NOS_FINAL) }
#undef L_
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

static size_t al256(size_t n) { return (n + 255) & ~(size_t)255; }

extern "C" size_t sgpo_nos_workspace_bytes(const sgpo_dit* m, int B, int L) {
  if (!m || B <= 0 || L <= 0) return 0;
  const size_t M = (size_t)B * L, H = m->cfg.hidden;
  // h | delta | G | p0[32] | logits[vocab]
  return 3 * al256(M * H * 4) + al256(M * 32 * 4) + al256(M * m->cfg.vocab * 4);
}

extern "C" int sgpo_nos_denoise(sgpo_dit* m, const int64_t* xt, const float* sigma, int n_sigma, int B, int L,
                                const float* w_cls, int num_nos_steps, float nos_step_size, float nos_stability_coef,
                                float move_chance_t, float move_chance_s, int kl_batch, int mask_index, int noise_mode,
                                const float* noise, uint64_t seed, uint64_t offset, uint64_t row_offset,
                                const float* hidden_in, int64_t* xs_out, float* probs_out, float* delta_out,
                                void* nos_workspace,
                                size_t nos_workspace_bytes, void* dit_workspace, size_t dit_workspace_bytes,
                                void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  SGPO_REQUIRE(m && xt && w_cls && xs_out && nos_workspace && dit_workspace, "nos_denoise: null argument");
  SGPO_REQUIRE(B > 0 && L > 0 && num_nos_steps >= 0 && kl_batch > 0, "nos_denoise: B=%d L=%d steps=%d kl_batch=%d", B, L, num_nos_steps, kl_batch);
  SGPO_REQUIRE(m->cfg.vocab <= 32 && mask_index >= 0 && mask_index < m->cfg.vocab, "nos_denoise: vocab %d (<= 32), mask %d", m->cfg.vocab, mask_index);
  SGPO_REQUIRE(m->cfg.hidden == 512 || m->cfg.hidden == 128 || m->cfg.hidden == 256, "nos_denoise: hidden %d unsupported (128/256/512)", m->cfg.hidden);
  SGPO_REQUIRE(move_chance_t > 0.f && move_chance_s >= 0.f && move_chance_s < move_chance_t, "nos_denoise: need 0 <= move_chance_s < move_chance_t");
  SGPO_REQUIRE(((uintptr_t)nos_workspace & 255) == 0, "nos_denoise: workspace must be 256-B aligned");
  if (nos_workspace_bytes < sgpo_nos_workspace_bytes(m, B, L)) {
    sgpo_set_error("nos_denoise: workspace %zu < required %zu bytes", nos_workspace_bytes, sgpo_nos_workspace_bytes(m, B, L));
    return SGPO_ERR_WORKSPACE;
  }
  const size_t M = (size_t)B * L, H = m->cfg.hidden, V = m->cfg.vocab;
  char* base = (char*)nos_workspace;
  float* h = (float*)base; base += al256(M * H * 4);
  float* delta = (float*)base; base += al256(M * H * 4);
  float* G = (float*)base; base += al256(M * H * 4);
  float* p0 = (float*)base; base += al256(M * 32 * 4);
  float* logits = (float*)base;
  // 1. backbone once: hidden state + adaLN rows of the final layer (left in the DiT workspace)
  //    (hidden_in given: skip the blocks, only the conditioning + adaLN rows are computed)
  int rc = sgpo_dit_forward(m, xt, sigma, n_sigma, B, L, hidden_in, logits, h, dit_workspace, dit_workspace_bytes, stream_);
  if (rc) return rc;
  const float* fmod = dit_ws_final_mod(m, B, L, dit_workspace);
  const int ns = sigma ? n_sigma : 1;
  NosArgs a;
  a.h = h; a.delta = delta; a.G = G; a.p0 = p0; a.logits_out = logits; a.xt = xt;
  a.W = m->out_lin_f32; a.bias = m->out_lin_b; a.norm_w = m->out_norm; a.shift = fmod; a.scale = fmod + H;
  a.w_cls = w_cls; a.mod_stride = (ns == 1) ? 0 : m->n_mod; a.rows = (int)M; a.L = L; a.B = B; a.V = (int)V;
  a.mask_index = mask_index;
  const float r = move_chance_s / move_chance_t;
  a.cscale = nos_stability_coef * (1.0f - r) / (float)kl_batch;
  a.lr = nos_step_size; a.inv_L = 1.0f / (float)L;
  auto launch = [&](int mode) {
    ProfScope prof_scope(PK_MISC, st, (double)M * H * 20.0);
    if (H == 512) return launch_nos<512>(mode, a, st);
    if (H == 256) return launch_nos<256>(mode, a, st);
    return launch_nos<128>(mode, a, st);
  };
  rc = launch(NOS_INIT);
  if (rc) return rc;
  for (int s = 0; s < num_nos_steps; ++s) {
    rc = launch(NOS_STEP);
    if (rc) return rc;
  }
  rc = launch(NOS_FINAL);
  if (rc) return rc;
  if (delta_out) SGPO_CHECK_HIP(hipMemcpyAsync(delta_out, delta, M * H * 4, hipMemcpyDeviceToDevice, st));
  // 2. guided_probs = exp(log p + log(1 - r)), mask column r; _sample_categorical; carry-over
  return sgpo_reverse_step_mdlm(logits, xt, nullptr, nullptr, L, B, L, (int)V, mask_index, 1.0f, r, 0.f, SGPO_STEP_HARD,
                                noise_mode, noise, seed, offset, row_offset, xs_out, probs_out, nullptr, stream_);
}
