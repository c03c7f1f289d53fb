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
// HBM per row: h read once (4 H bytes), logits written once; delta and G stay in registers over all inner steps
// (one launch); W (vocab x H fp32) in LDS.
#include "../../include/sgpo_hip.h"
#include "common.h"
#include "dit_model.h"
#include "prof.h"

const float* dit_ws_final_mod(const sgpo_dit* m, int B, int L, void* workspace);
const float* dit_ws_hidden(const sgpo_dit* m, int B, int L, void* workspace);

#ifndef NOS_THREADS
#define NOS_THREADS 768   // 12 waves share one copy of W in LDS: 3 per SIMD at <= 168 VGPRs
#endif
#define NOS_QUEUES 32     // work counters, 256 B apart
#ifndef NOS_KGROUP
#define NOS_KGROUP 4   // W rows whose LDS reads may be in flight together
#endif

struct NosArgs {
  const float* h; float* delta_out; float* logits_out;
  int* next_pair;   // NOS_QUEUES counters, 64 ints apart, zeroed before the launch
  const int64_t* xt;
  const float *W, *bias, *norm_w, *shift, *scale, *w_cls;
  int mod_stride, rows, L, B, V, mask_index, n_steps;
  float cscale, lr, inv_L;
};

// A wave works on TWO rows at a time, packed in the two halves of 64-bit register pairs (v_pk_fma_f32 & co.): every W
// element read from LDS feeds both rows (at one row per wave the 2 x 32 x H ds_read_b32 per row and step bound the
// kernel), and every FMA instruction does both rows' work.  Per row the arithmetic and its order are those of one row.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f v2(float a) { return v2f{a, a}; }
__device__ __forceinline__ v2f xor2(v2f v, int m) { return v2f{__shfl_xor(v.x, m), __shfl_xor(v.y, m)}; }
__device__ __forceinline__ v2f wave_sum2(v2f v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += xor2(v, o);
  return v;
}
__device__ __forceinline__ v2f wave_max2(v2f v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const v2f w = xor2(v, o); v = v2f{fmaxf(v.x, w.x), fmaxf(v.y, w.y)}; }
  return v;
}

// Logits as a reduce-scatter: every lane forms its partial dot product of all 32 W rows, and 31 exchanges (instead of
// 6 per logit) leave lane l with the total of logit k = (l >> 1) & 31 up to the last lane pair.  Evaluated depth first:
//   T<32>(j) = partial dot of row j;   T<S>(j) = (b ? v : u) + xor_exchange(b ? u : v, 2S),  u = T<2S>(j), v = T<2S>(j + S),
// b = lane & 2S, so at most one partial per level is live (5 register pairs instead of the 32 accumulators of the
// breadth-first form; same pairs, same order of additions).  Rows V..31 of the LDS image are zero: no `k < V` branch,
// the whole sweep is one basic block and the LDS reads of the next rows run under the FMA chains.
template <int H, int S>
__device__ __forceinline__ v2f nos_tree(const float* wl, int lane, const v2f (&xn)[H / 64], int j) {
  if constexpr (S == 32) {
    v2f out = v2(0.f);
#pragma unroll
    for (int i = 0; i < H / 64; ++i) out = __builtin_elementwise_fma(v2(wl[j * H + i * 64 + lane]), xn[i], out);
    return out;
  } else {
    const v2f u = nos_tree<H, 2 * S>(wl, lane, xn, j);
    const v2f v = nos_tree<H, 2 * S>(wl, lane, xn, j + S);
    if constexpr (S == 16 || S == 8) {
      // lanes with bit 2S clear need u of (lane, lane ^ 2S), the others v: v_permlane32_swap / v_permlane16_swap move
      // exactly those halves (one VALU instruction per row instead of two selects and a trip through the LDS crossbar)
      v2f lo, hi;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const auto r = S == 16 ? __builtin_amdgcn_permlane32_swap(__float_as_uint(u[c]), __float_as_uint(v[c]), false, false)
                               : __builtin_amdgcn_permlane16_swap(__float_as_uint(u[c]), __float_as_uint(v[c]), false, false);
        lo[c] = __uint_as_float(r[0]);
        hi[c] = __uint_as_float(r[1]);
      }
      return lo + hi;
    } else {
      const bool b = lane & (2 * S);
      return (b ? v : u) + xor2(b ? u : v, 2 * S);
    }
  }
}

// final layer for the two rows: x -> y = LN(x), rstd, and the logit of column kl = (lane >> 1) & 31 (both lanes of a
// pair hold it; -inf for kl >= V).  g = norm_w * (1 + scale) and shift are per row (its sequence's adaLN rows).
template <int H>
__device__ __forceinline__ v2f nos_forward(const NosArgs& a, const float* wl, int lane, const v2f (&x)[H / 64],
                                           const float (&nw)[H / 64], const float* const (&sh)[2],
                                           const float* const (&sc)[2], v2f (&y)[H / 64], v2f& rstd) {
  constexpr int PER = H / 64;
  v2f s = v2(0.f);
#pragma unroll
  for (int i = 0; i < PER; ++i) s += x[i];
  const v2f mean = wave_sum2(s) * (1.0f / H);
  v2f q = v2(0.f);
#pragma unroll
  for (int i = 0; i < PER; ++i) { const v2f d = x[i] - mean; q += d * d; }
  const v2f var = wave_sum2(q) * (1.0f / H) + 1e-5f;
  rstd = v2f{rsqrtf(var.x), rsqrtf(var.y)};
  v2f xn[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int col = i * 64 + lane;
    y[i] = (x[i] - mean) * rstd;
    xn[i] = y[i] * (nw[i] * (1.f + v2f{sc[0][col], sc[1][col]})) + v2f{sh[0][col], sh[1][col]};
  }
  const v2f t = nos_tree<H, 1>(wl, lane, xn, 0);
  const v2f z = t + xor2(t, 1);
  const int kl = (lane >> 1) & 31;
  return kl < a.V ? z + a.bias[kl] : v2(-INFINITY);
}

// softmax over the non-mask columns (subs parameterisation: mask logit += -1e6 -> weight 0); lane pair kl holds p[kl]
__device__ __forceinline__ v2f nos_softmax(v2f zl, bool valid, bool first) {
  const v2f zmax = wave_max2(valid ? zl : v2(-INFINITY));
  const v2f e = valid ? v2f{expf(zl.x - zmax.x), expf(zl.y - zmax.y)} : v2(0.f);
  const v2f sum = wave_sum2(first ? e : v2(0.f));
  return e * v2f{1.0f / sum.x, 1.0f / sum.y};
}

// grad = dx - w_cls / L: the product is rounded before the subtraction (as `0 - c` needs it for unmasked rows)
__device__ __forceinline__ v2f nos_grad(v2f dx, float w_cls, float inv_L) {
#pragma clang fp contract(off)
  const float c = w_cls * inv_L;
  return dx - c;
}

// Adagrad on one element pair: G += grad^2, delta -= lr grad / (sqrt(G) + 1e-10).  v_sqrt_f32 and v_rcp_f32 (1 ulp
// each) instead of the correctly rounded expansions, which are ~25 instructions per element and made the update as
// expensive as a sweep over W; the relative error per step (~3e-7) is far inside the test tolerance on delta (2e-3).
// (No contraction in here: the same operations wherever it is inlined, so an unmasked row gets the same bits from the
// per-column closed form as from riding along in a pair with a masked row.)
__device__ __forceinline__ void nos_adagrad(v2f& G, v2f& delta, v2f grad, float lr) {
#pragma clang fp contract(off)
  G = __builtin_elementwise_fma(grad, grad, G);
  const v2f den = v2f{__builtin_amdgcn_sqrtf(G.x), __builtin_amdgcn_sqrtf(G.y)} + 1e-10f;
  const v2f step = (lr * grad) * v2f{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  delta = delta - step;
}

// The whole NOS optimisation of a row in ONE launch: rows are independent across inner steps, so delta and the
// Adagrad accumulator G stay in registers from the first step to the last and never touch HBM (the per-step version
// moved 20 H bytes per row and step and was bound by that).  The waves of a workgroup share W in LDS; a wave draws
// row pairs from counters (a pair with masked rows costs ~12 sweeps of W, one without 1: a fixed deal leaves waves idle).
//   p0 = softmax(logits(h));  n_steps x { z(h + delta) -> dz -> closed-form backward -> Adagrad };  logits(h + delta)
// Unmasked rows: dL/dz is 0 there, so the gradient is the constant -w_cls / L and delta follows a recurrence that
// is the same for every such row (dU, once per wave; same bits as the full computation: dx = rstd * (0 - 0 - y * 0)
// = 0, 0 - c == -c); a pair with one masked row runs the full computation for both with dz = 0 on the unmasked one.
template <int H>
__global__ __launch_bounds__(NOS_THREADS) void nos_fused_kernel(NosArgs a) {
  extern __shared__ float wl[];  // [32][H]: W, rows V..31 zero
  constexpr int PER = H / 64;
  for (int i = threadIdx.x; i < 32 * H; i += NOS_THREADS) wl[i] = i < a.V * H ? a.W[i] : 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int kl = (lane >> 1) & 31;
  const bool first = (lane & 1) == 0;
  const bool valid = kl < a.V && kl != a.mask_index;
  float nw[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    nw[i] = a.norm_w[i * 64 + lane];
  }
  // an unmasked row's gradient is the constant -w_cls / L at every step, so its delta after n_steps depends on the
  // column only: run that recurrence once per wave (the same nos_adagrad, the same bits as the row-by-row loop)
  float dU[PER];
  {
    v2f Gu[PER], du[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) { Gu[i] = v2(0.f); du[i] = v2(0.f); }
    for (int s = 0; s < a.n_steps; ++s)
#pragma unroll
      for (int i = 0; i < PER; ++i) nos_adagrad(Gu[i], du[i], nos_grad(v2(0.f), a.w_cls[i * 64 + lane], a.inv_L), a.lr);
#pragma unroll
    for (int i = 0; i < PER; ++i) dU[i] = du[i].x;
  }
  // Row pairs are drawn from NOS_QUEUES counters (one cache line each), not dealt: a pair with a masked row costs 12
  // sweeps of W, one without costs 1.  A single counter would serialise ~39 k same-address atomics (~10 ns each, 0.4 ms:
  // more than the sweeps of a mostly unmasked batch); each wave starts on its own queue and moves on when one is empty.
  const int n_pairs = (a.rows + 1) / 2;
  const int per_q = (n_pairs + NOS_QUEUES - 1) / NOS_QUEUES;
  int q = (blockIdx.x * (NOS_THREADS / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) % NOS_QUEUES;
  int empty_seen = 0;
  while (empty_seen < NOS_QUEUES) {   // every wave leaves after at most NOS_QUEUES failed draws: the counters only grow
    int idx = 0;
    if (lane == 0) idx = atomicAdd(a.next_pair + q * 64, 1);
    idx = __builtin_amdgcn_readfirstlane(idx);      // scalar: row numbers and row base addresses stay in SGPRs
    const int pair = q * per_q + idx;
    if (idx >= per_q || pair >= n_pairs) {
      q = q + 1 == NOS_QUEUES ? 0 : q + 1;
      ++empty_seen;
      continue;
    }
    int rows[2];
    bool live[2], masked[2];
    const float* sh[2];
    const float* sc[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      live[r] = 2 * pair + r < a.rows;
      rows[r] = live[r] ? 2 * pair + r : a.rows - 1;   // an odd tail computes its last row twice and stores it once
      masked[r] = live[r] && (int)a.xt[rows[r]] == a.mask_index;
      int b = rows[r] / a.L;
      if (b > a.B - 1) b = a.B - 1;
      sh[r] = a.shift + (size_t)b * a.mod_stride;
      sc[r] = a.scale + (size_t)b * a.mod_stride;
    }
    const bool any = masked[0] || masked[1];
    const float* h0 = a.h + (size_t)rows[0] * H + lane;
    const float* h1 = a.h + (size_t)rows[1] * H + lane;
    v2f delta[PER], G[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) { delta[i] = v2(0.f); G[i] = v2(0.f); }
    v2f y[PER], rstd, p0 = v2(0.f);
    // h is read again at every step (2 KB per row, from L2) rather than held in 2 x PER registers across the loop
    if (any && a.n_steps > 0) {
      v2f x[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) x[i] = v2f{h0[i * 64], h1[i * 64]};
      p0 = nos_softmax(nos_forward<H>(a, wl, lane, x, nw, sh, sc, y, rstd), valid, first);
    }
    if (!any) {
#pragma unroll
      for (int i = 0; i < PER; ++i) delta[i] = v2(dU[i]);
    }
    for (int s = 0; any && s < a.n_steps; ++s) {
      v2f dx[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) dx[i] = v2(0.f);
      {
        v2f x[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) x[i] = v2f{h0[i * 64], h1[i * 64]} + delta[i];
        const v2f pl = nos_softmax(nos_forward<H>(a, wl, lane, x, nw, sh, sc, y, rstd), valid, first);
        v2f dzl = a.cscale * (pl - p0);
        dzl = v2f{valid && masked[0] ? dzl.x : 0.f, valid && masked[1] ? dzl.y : 0.f};
        // closed-form backward: dxn = W^T dz, dy = dxn * g, dx = rstd * (dy - mean(dy) - y * mean(dy * y))
        // (dz is 0 for the mask column and for k >= V, where the LDS rows are zero too: fma(w, 0, dx) == dx, no branch)
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          const v2f dz = v2f{__int_as_float(__builtin_amdgcn_readlane(__float_as_int(dzl.x), 2 * k)),
                             __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dzl.y), 2 * k))};   // wave-uniform: SGPRs
#pragma unroll
          for (int i = 0; i < PER; ++i) dx[i] = __builtin_elementwise_fma(v2(wl[k * H + i * 64 + lane]), dz, dx[i]);
          if ((k & (NOS_KGROUP - 1)) == NOS_KGROUP - 1) __builtin_amdgcn_sched_barrier(0);   // bound the read-ahead: registers
        }
        v2f s1 = v2(0.f), s2 = v2(0.f);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const int col = i * 64 + lane;
          dx[i] *= nw[i] * (1.f + v2f{sc[0][col], sc[1][col]});
          s1 += dx[i];
          s2 += dx[i] * y[i];
        }
        const v2f m1 = wave_sum2(s1) * (1.0f / H), m2 = wave_sum2(s2) * (1.0f / H);
#pragma unroll
        for (int i = 0; i < PER; ++i) dx[i] = rstd * (dx[i] - m1 - y[i] * m2);
      }
      // grad = dx - w_cls / L (the mean-pool + Linear head), then Adagrad
#pragma unroll
      for (int i = 0; i < PER; ++i) nos_adagrad(G[i], delta[i], nos_grad(dx[i], a.w_cls[i * 64 + lane], a.inv_L), a.lr);   // (w_cls from L1: 8 registers fewer)
    }
    // the final logits are formed for every row
    v2f zl;
    {
      v2f x[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) x[i] = v2f{h0[i * 64], h1[i * 64]} + delta[i];
      zl = nos_forward<H>(a, wl, lane, x, nw, sh, sc, y, rstd);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (!live[r]) continue;
      if (first && kl < a.V) a.logits_out[(size_t)rows[r] * a.V + kl] = r ? zl.y : zl.x;
      if (a.delta_out) {
#pragma unroll
        for (int i = 0; i < PER; ++i) a.delta_out[(size_t)rows[r] * H + i * 64 + lane] = r ? delta[i].y : delta[i].x;
      }
    }
  }
}

template <int H>
static int launch_nos(const NosArgs& a, hipStream_t st) {
  const size_t lds = (size_t)32 * H * sizeof(float);
  static int cap = 0;   // one round of resident workgroups: CUs x what the registers and LDS allow
  if (!cap) {
    SGPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nos_fused_kernel<H>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 32 * H * (int)sizeof(float)));
    int dev = 0, n_cu = 0, per_cu = 0;
    SGPO_CHECK_HIP(hipGetDevice(&dev));
    SGPO_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    SGPO_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nos_fused_kernel<H>, NOS_THREADS, 32 * H * sizeof(float)));
    cap = n_cu * (per_cu > 0 ? per_cu : 1);
  }
  const int want = (a.rows + 2 * (NOS_THREADS / 64) - 1) / (2 * (NOS_THREADS / 64));   // a wave takes rows in pairs
  SGPO_CHECK_HIP(hipMemsetAsync(a.next_pair, 0, NOS_QUEUES * 256, st));
  hipLaunchKernelGGL(nos_fused_kernel<H>, dim3(want < cap ? want : cap), dim3(NOS_THREADS), lds, st, a);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

static size_t al256(size_t n) { return (n + 255) & ~(size_t)255; }

extern "C" size_t sgpo_nos_workspace_bytes(const sgpo_dit* m, int B, int L) {
  if (!m || B <= 0 || L <= 0) return 0;
  const size_t M = (size_t)B * L;
  // logits[vocab] | the row-pair counters   (the hidden state is read where the forward left it; delta, G and p0
  // live in registers)
  return al256(M * m->cfg.vocab * 4) + NOS_QUEUES * 256;
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
  float* logits = (float*)base; base += al256(M * V * 4);
  int* next_pair = (int*)base;
  // 1. backbone once: hidden state + adaLN rows of the final layer (left in the DiT workspace)
  //    (hidden_in given: skip the blocks, only the conditioning + adaLN rows are computed)
  int rc = sgpo_dit_forward(m, xt, sigma, n_sigma, B, L, hidden_in, logits, nullptr, dit_workspace, dit_workspace_bytes, stream_);
  if (rc) return rc;
  const float* h = hidden_in ? hidden_in : dit_ws_hidden(m, B, L, dit_workspace);   // no copy: the kernel below runs before the workspace is used again
  const float* fmod = dit_ws_final_mod(m, B, L, dit_workspace);
  const int ns = sigma ? n_sigma : 1;
  NosArgs a;
  a.h = h; a.delta_out = delta_out; a.logits_out = logits; a.xt = xt; a.next_pair = next_pair;
  a.W = m->out_lin_f32; a.bias = m->out_lin_b; a.norm_w = m->out_norm; a.shift = fmod; a.scale = fmod + H;
  a.w_cls = w_cls; a.mod_stride = (ns == 1) ? 0 : m->n_mod; a.rows = (int)M; a.L = L; a.B = B; a.V = (int)V;
  a.mask_index = mask_index; a.n_steps = num_nos_steps;
  const float r = move_chance_s / move_chance_t;
  a.cscale = nos_stability_coef * (1.0f - r) / (float)kl_batch;
  a.lr = nos_step_size; a.inv_L = 1.0f / (float)L;
  {
    // h read once, logits (and delta if asked for) written once; the flop side: (2 + 2 n) forward and n backward sweeps of V x H
    ProfScope prof_scope(PK_MISC, st, (double)M * (H * (delta_out ? 8.0 : 4.0) + V * 4.0));
    rc = H == 512 ? launch_nos<512>(a, st) : H == 256 ? launch_nos<256>(a, st) : launch_nos<128>(a, st);
    if (rc) return rc;
  }
  // 2. guided_probs = exp(log p + log(1 - r)), mask column r; _sample_categorical; carry-over
  return sgpo_reverse_step_mdlm(logits, xt, nullptr, nullptr, L, B, L, (int)V, mask_index, 1.0f, r, 0.f, SGPO_STEP_HARD,
                                noise_mode, noise, seed, offset, row_offset, xs_out, probs_out, nullptr, stream_);
}
