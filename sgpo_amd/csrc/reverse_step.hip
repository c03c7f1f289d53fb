// Fused reverse-step kernels (HBM-bound): logits row -> parameterisation -> reverse
// probabilities -> guidance reweighting -> categorical sample.  One vocabulary row per
// 32-lane half-wave (K <= 32), lane k holds class k; all row reductions are half-wave
// shuffles; logits are read once, coalesced, straight into registers.
//
// Reference algebra (bug-compatible forms):
//   subs parameterisation  mdlm_diffusion.py:325-341
//   MDLM.p_sample          models/mdlm.py:55-98
//   CG reweight + sample   sampling/cls_guidance.py:38-52, models/mdlm.py:399-409
//   _sample_categorical    mdlm_diffusion.py:69-73
//   UDLM posterior         mdlm_diffusion.py:392-419
//   q_xt                   mdlm_diffusion.py:585-597
#include "../../include/sgpo_hip.h"
#include "common.h"
#include "prof.h"
#include "mlp.h"

#define NEG_INF_REF (-1000000.0f)

// index of the maximum inside a 32-lane half; ties -> lowest index (torch.argmax on CPU)
__device__ __forceinline__ int half_argmax(float v, int k) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    float ov = __shfl_xor(v, o);
    int ok = __shfl_xor(k, o);
    if (ov > v || (ov == v && ok < k)) { v = ov; k = ok; }
  }
  return k;
}

// as noise_denominator, the Philox word already drawn
__device__ __forceinline__ float noise_denominator_w(int noise_mode, const float* noise, size_t local_idx, uint32_t word,
                                                     bool multinomial) {
  if (noise_mode == SGPO_NOISE_NONE) return 1.0f;
  if (noise_mode == SGPO_NOISE_EXPONENTIAL) return noise[local_idx];
  float u;
  if (noise_mode == SGPO_NOISE_UNIFORM) {
    u = noise[local_idx];
    if (multinomial) return -logf(1.0f - u);
    return 1e-10f - logf(u + 1e-10f);
  }
  u = u01_open(word);
  if (multinomial) return -logf(u);
  return 1e-10f - logf(u + 1e-10f);
}

// Half-wave (32-lane) all-reduces for the MDLM kernel: four DPP steps (lane^1, lane^2, mirror inside 8, mirror inside 16:
// one VALU instruction each, no LDS crossbar) and one swizzle across the two rows of 16, instead of five ds_bpermute.
#define SGPO_DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, true))
#define SGPO_SWZ16_F(v) __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F))   // lane ^ 16 within 32
__device__ __forceinline__ float hsum32(float v) {
  v += SGPO_DPP_F(v, 0xB1);
  v += SGPO_DPP_F(v, 0x4E);
  v += SGPO_DPP_F(v, 0x141);
  v += SGPO_DPP_F(v, 0x140);
  return v + SGPO_SWZ16_F(v);
}
__device__ __forceinline__ float hmax32(float v) {
  v = fmaxf(v, SGPO_DPP_F(v, 0xB1));
  v = fmaxf(v, SGPO_DPP_F(v, 0x4E));
  v = fmaxf(v, SGPO_DPP_F(v, 0x141));
  v = fmaxf(v, SGPO_DPP_F(v, 0x140));
  return fmaxf(v, SGPO_SWZ16_F(v));
}
// lowest lane of the half holding the maximum (torch.argmax on CPU); no lane matches (every value NaN) -> 0, which is
// what the exchange loop of half_argmax leaves in lane 0
__device__ __forceinline__ int hargmax32(float v) {
  const float m = hmax32(v);
  const unsigned long long bal = __builtin_amdgcn_ballot_w64(v == m);
  const unsigned hm = (unsigned)(bal >> (threadIdx.x & 32));
  return hm ? __ffs(hm) - 1 : 0;
}

__device__ __forceinline__ float noise_denominator(int noise_mode, const float* noise, size_t local_idx,
                                                   uint64_t seed, uint64_t offset, uint64_t global_idx, bool multinomial) {
  // returns the value each probability is divided by before the argmax
  if (noise_mode == SGPO_NOISE_NONE) return 1.0f;
  if (noise_mode == SGPO_NOISE_EXPONENTIAL) return noise[local_idx];
  float u;
  if (noise_mode == SGPO_NOISE_UNIFORM) {
    u = noise[local_idx];
    if (multinomial) return -logf(1.0f - u);  // exponential from a uniform in [0,1)
    return 1e-10f - logf(u + 1e-10f);
  }
  Philox4 r = philox4x32_10(seed, global_idx, offset);
  u = u01_open(r.x);
  if (multinomial) return -logf(u);
  return 1e-10f - logf(u + 1e-10f);
}

// One row of the step, its inputs already in registers (z: the logit of class k or -inf; lr: the jump log-ratio of
// class k, used when has_lr).  Called with all 32 lanes of the half-wave.
__device__ __forceinline__ void reverse_step_mdlm_row(
    int row, int k, bool act, int xt, float z, bool guided, bool has_lr, float lr, int K, int mask_index, float mt, float ms,
    float guide_temp, int step_mode, int noise_mode, const float* __restrict__ noise, uint32_t philox_word,
    int64_t* __restrict__ x_out, float* __restrict__ probs_out, int* __restrict__ nan_flag) {
  // subs parameterisation
  if (k == mask_index) z += NEG_INF_REF;
  const float zmax = hmax32(z);
  const float se = hsum32(act ? expf(z - zmax) : 0.f);
  float lp = z - (zmax + logf(se));  // logits - logsumexp
  if (xt != mask_index) lp = (k == xt) ? 0.f : NEG_INF_REF;
  float q = expf(lp) * (mt - ms);
  if (k == mask_index) q = ms;
  if (!act) q = 0.f;

  if (step_mode == SGPO_STEP_SOFT) {
    if (probs_out && act) probs_out[(size_t)row * K + k] = q;
    return;
  }
  const size_t lidx = (size_t)row * K + k;
  if (step_mode == SGPO_STEP_HARD) {
    // _sample_categorical + carry-over (models/mdlm.py:81-84)
    float den = act ? noise_denominator_w(noise_mode, noise, lidx, philox_word, false) : 1.f;
    float v = act ? q / den : -INFINITY;
    int samp = hargmax32(v);
    if (probs_out && act) probs_out[lidx] = q;
    if (k == 0) x_out[row] = (xt != mask_index) ? xt : samp;
    return;
  }
  // classifier guidance (cls_guidance.py:38-52)
  if (k == xt) q -= 1.0f;
  if (has_lr) {
    lr = lr / guide_temp;
    if (lr == lr) lr = fminf(lr, 80.0f);  // torch.clamp propagates NaN, fminf would drop it
    q = q * expf(lr);
  }
  if (guided && nan_flag && act && q != q) atomicOr(nan_flag, 1);   // models/mdlm.py:410-411
  if (k == xt) q += 1.0f;
  q = q < 0.f ? 0.f : q;   // torch.clamp(min=0) keeps a NaN (fmaxf would turn it into 0 and hide it from the check below)
  if (!act) q = 0.f;
  const float tot = hsum32(q);
  // torch.multinomial refuses a row whose sum is 0, inf or NaN (cls_guidance.py:52); the kernel would draw token 0
  if (nan_flag && k == 0 && !(tot > 0.f && tot <= 3.4028234e38f)) atomicOr(nan_flag, 2);
  const float p = q / tot;
  float den = act ? noise_denominator_w(noise_mode, noise, lidx, philox_word, true) : 1.f;
  float v = act ? p / den : -INFINITY;
  int samp = hargmax32(v);
  if (probs_out && act) probs_out[lidx] = p;
  if (k == 0) x_out[row] = samp;
}

// A half-wave takes RS_ROWS consecutive rows and requests all their inputs before the first row is worked on: with
// one 124-byte row per half-wave the kernel had ~0.5 KB in flight per wave and was bound by memory latency
// (1 TB/s with every wave slot of the chip occupied).
#define RS_ROWS 4
__global__ __launch_bounds__(256) void reverse_step_mdlm_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ x, const float* __restrict__ log_ratio,
    const int* __restrict__ slot_of_pos, int n_slots, int rows, int L, int K, int mask_index, float mt, float ms,
    float guide_temp, int step_mode, int noise_mode, const float* __restrict__ noise, uint64_t seed,
    uint64_t offset, uint64_t row_offset, int64_t* __restrict__ x_out, float* __restrict__ probs_out,
    int* __restrict__ nan_flag, RsGroups gs, const uint64_t* __restrict__ key_dev, int key_g0) {
  if (gs.n > 0) {   // grouped launch: blockIdx.y is a draw of the stacked batch, with its own rows and Philox key
    const RsGroup& G = gs.g[blockIdx.y];
    const size_t r0 = (size_t)G.seq_lo * L;
    logits += r0 * K; x += r0; x_out += r0;
    log_ratio = G.guided ? log_ratio + (size_t)G.seq_lo * n_slots * K : nullptr;
    rows = G.n_seq * L; seed = G.seed; offset = G.offset; row_offset = G.row_offset;
  }
  // Philox key from DEVICE memory (the *_devkey entry points): a launch recorded in a HIP graph is replayed with a new
  // key without touching its kernel arguments.  Per draw 4 words: seed, first global row, call-offset base, unused.
  if (key_dev) {
    const uint64_t* kd = key_dev + 4 * (size_t)(key_g0 + (gs.n > 0 ? (int)blockIdx.y : 0));
    seed = kd[0]; row_offset = kd[1]; offset += kd[2];
  }
  const int half = (blockIdx.x * 256 + threadIdx.x) >> 5;
  const int k = threadIdx.x & 31;
  const int row0 = half * RS_ROWS;
  if (row0 >= rows) return;
  const bool act = k < K;
  const bool guided = step_mode == SGPO_STEP_SOFT_CG && log_ratio != nullptr && guide_temp != 0.f;
  int xt[RS_ROWS];
  float z[RS_ROWS], lr[RS_ROWS];
  bool has_lr[RS_ROWS];
#pragma unroll
  for (int r = 0; r < RS_ROWS; ++r) {
    const int row = row0 + r;
    const bool valid = row < rows;          // uniform over the half-wave
    xt[r] = valid ? (int)x[row] : mask_index;
    z[r] = (valid && act) ? logits[(size_t)row * K + k] : -INFINITY;
    lr[r] = 0.f;
    has_lr[r] = false;
    if (guided && valid && act) {
      const int l = row % L;
      const int slot = slot_of_pos ? slot_of_pos[l] : l;
      if (slot >= 0) {
        lr[r] = log_ratio[((size_t)(row / L) * n_slots + slot) * K + k];
        has_lr[r] = true;
      }
    }
  }
  // In-kernel noise: one Philox4x32-10 call serves class k of FOUR consecutive global rows (counter = (global row >> 2)
  // * K + k, word = global row & 3), so a call is shared by the rows of this half-wave instead of one word of four
  // being used.  Keyed by the global row, so any sharding draws the same numbers; a group that straddles two
  // counters (first global row no multiple of 4) makes a second call.
  uint32_t pw[RS_ROWS] = {0, 0, 0, 0};
  if (noise_mode == SGPO_NOISE_PHILOX && step_mode != SGPO_STEP_SOFT) {
    const uint64_t g0 = (uint64_t)row0 + row_offset;
    const Philox4 a = philox4x32_10(seed, (g0 >> 2) * (uint64_t)K + k, offset);
    Philox4 b = a;
    if (g0 & 3) b = philox4x32_10(seed, ((g0 >> 2) + 1) * (uint64_t)K + k, offset);
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) {
      const uint64_t g = g0 + r;
      const Philox4& c = ((g >> 2) == (g0 >> 2)) ? a : b;
      const int wsel = (int)(g & 3);
      pw[r] = wsel == 0 ? c.x : wsel == 1 ? c.y : wsel == 2 ? c.z : c.w;
    }
  }
#pragma unroll
  for (int r = 0; r < RS_ROWS; ++r) {
    if (row0 + r < rows)
      reverse_step_mdlm_row(row0 + r, k, act, xt[r], z[r], guided, has_lr[r], lr[r], K, mask_index, mt, ms, guide_temp, step_mode,
                            noise_mode, noise, pw[r], x_out, probs_out, nan_flag);
  }
}

static int reverse_step_mdlm_impl(const float* logits, const int64_t* x, const float* log_ratio,
                                  const int* slot_of_pos, int n_slots, int B, int L, int K, int mask_index,
                                  float move_chance_t, float move_chance_s, float guide_temp, int step_mode,
                                  int noise_mode, const float* noise, uint64_t seed, uint64_t offset,
                                  uint64_t row_offset, int64_t* x_out, float* probs_out, int* nan_flag,
                                  const uint64_t* key_dev, void* stream) {
  SGPO_REQUIRE(logits && x, "reverse_step_mdlm: null logits/x");
  SGPO_REQUIRE(B > 0 && L > 0 && K > 0 && K <= 32, "reverse_step_mdlm: B=%d L=%d K=%d (K <= 32)", B, L, K);
  SGPO_REQUIRE(mask_index >= 0 && mask_index < K, "reverse_step_mdlm: mask_index %d outside [0,%d)", mask_index, K);
  SGPO_REQUIRE(step_mode >= 0 && step_mode <= 2, "reverse_step_mdlm: bad step_mode %d", step_mode);
  SGPO_REQUIRE(noise_mode >= 0 && noise_mode <= 3, "reverse_step_mdlm: bad noise_mode %d", noise_mode);
  SGPO_REQUIRE(step_mode == SGPO_STEP_SOFT || x_out, "reverse_step_mdlm: x_out required");
  SGPO_REQUIRE(step_mode != SGPO_STEP_SOFT || probs_out, "reverse_step_mdlm: probs_out required for SOFT");
  SGPO_REQUIRE(noise_mode == SGPO_NOISE_PHILOX || noise_mode == SGPO_NOISE_NONE || step_mode == SGPO_STEP_SOFT || noise, "reverse_step_mdlm: noise tensor missing");
  SGPO_REQUIRE(log_ratio == nullptr || slot_of_pos != nullptr || n_slots == L, "reverse_step_mdlm: n_slots must equal L without a slot map");
  const int rows = B * L;
  const int blocks = ((rows + RS_ROWS - 1) / RS_ROWS + 7) / 8;   // 8 half-waves per block, RS_ROWS rows each
  // algorithmic bytes: logits + x in, x out (+ ratio for guided steps)
  ProfScope prof_scope(PK_REVERSE_STEP, (hipStream_t)stream,
                       (double)rows * (K * 4.0 * ((log_ratio && guide_temp != 0.f && step_mode == SGPO_STEP_SOFT_CG) ? 2 : 1) + 16.0));
  hipLaunchKernelGGL(reverse_step_mdlm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, x, log_ratio,
                     slot_of_pos, n_slots, rows, L, K, mask_index, move_chance_t, move_chance_s, guide_temp, step_mode,
                     noise_mode, noise, seed, offset, row_offset, x_out, probs_out, nan_flag, RsGroups{}, key_dev, 0);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

extern "C" int sgpo_reverse_step_mdlm(const float* logits, const int64_t* x, const float* log_ratio,
                                      const int* slot_of_pos, int n_slots, int B, int L, int K, int mask_index,
                                      float move_chance_t, float move_chance_s, float guide_temp, int step_mode,
                                      int noise_mode, const float* noise, uint64_t seed, uint64_t offset,
                                      uint64_t row_offset, int64_t* x_out, float* probs_out, int* nan_flag,
                                      void* stream) {
  return reverse_step_mdlm_impl(logits, x, log_ratio, slot_of_pos, n_slots, B, L, K, mask_index, move_chance_t, move_chance_s,
                                guide_temp, step_mode, noise_mode, noise, seed, offset, row_offset, x_out, probs_out, nan_flag,
                                nullptr, stream);
}

extern "C" int sgpo_reverse_step_mdlm_devkey(const float* logits, const int64_t* x, const float* log_ratio,
                                             const int* slot_of_pos, int n_slots, int B, int L, int K, int mask_index,
                                             float move_chance_t, float move_chance_s, float guide_temp, int step_mode,
                                             uint64_t offset, const uint64_t* key_dev, int64_t* x_out, float* probs_out,
                                             int* nan_flag, void* stream) {
  SGPO_REQUIRE(key_dev != nullptr, "reverse_step_mdlm_devkey: null key");
  return reverse_step_mdlm_impl(logits, x, log_ratio, slot_of_pos, n_slots, B, L, K, mask_index, move_chance_t, move_chance_s,
                                guide_temp, step_mode, SGPO_NOISE_PHILOX, nullptr, 0, offset, 0, x_out, probs_out, nan_flag,
                                key_dev, stream);
}

// the classifier-guidance step of several stacked draws in one launch (in-kernel Philox; see mlp.h)
int launch_reverse_step_mdlm_groups(const RsGroups& gs, int max_seq, const float* logits, const int64_t* x,
                                    const float* log_ratio, const int* slot_of_pos, int n_slots, int L, int K,
                                    int mask_index, float move_chance_t, float move_chance_s, float guide_temp,
                                    int64_t* x_out, int* nan_flag, hipStream_t stream, const uint64_t* key_dev, int key_g0) {
  SGPO_REQUIRE(gs.n > 0 && gs.n <= SGPO_MAX_GROUPS && logits && x && x_out, "reverse_step_mdlm_groups: bad arguments");
  SGPO_REQUIRE(L > 0 && K > 0 && K <= 32 && mask_index >= 0 && mask_index < K, "reverse_step_mdlm_groups: L=%d K=%d mask %d", L, K, mask_index);
  SGPO_REQUIRE(log_ratio == nullptr || slot_of_pos != nullptr || n_slots == L, "reverse_step_mdlm_groups: n_slots must equal L without a slot map");
  const int rows = max_seq * L;
  const int blocks = ((rows + RS_ROWS - 1) / RS_ROWS + 7) / 8;
  double total = 0;
  for (int g = 0; g < gs.n; ++g) total += (double)gs.g[g].n_seq * L * (K * 4.0 * ((gs.g[g].guided && guide_temp != 0.f) ? 2 : 1) + 16.0);
  ProfScope prof_scope(PK_REVERSE_STEP, stream, total);
  hipLaunchKernelGGL(reverse_step_mdlm_kernel, dim3(blocks, gs.n), dim3(256), 0, stream, logits, x, log_ratio, slot_of_pos,
                     n_slots, 0, L, K, mask_index, move_chance_t, move_chance_s, guide_temp, SGPO_STEP_SOFT_CG,
                     SGPO_NOISE_PHILOX, (const float*)nullptr, (uint64_t)0, (uint64_t)0, (uint64_t)0, x_out, (float*)nullptr,
                     nan_flag, gs, key_dev, key_g0);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

// ---------------------------------------------------------------------------
// UDLM: log-softmax (d3pm parameterisation, no mask column) then _compute_posterior (models/mdlm.py:87-95).
// SOFT_CG continues with the classifier-guidance algebra of sampling/cls_guidance.py:38-52 on the posterior -- the
// uniform-diffusion prior is non-zero in every column, so log_ratio holds ALL jump ratios (no zero-rate skipping).
__global__ __launch_bounds__(256) void reverse_step_udlm_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ x, const float* __restrict__ log_ratio,
    const int* __restrict__ slot_of_pos, int n_slots, int rows, int L, int K, float mt, float ms, float guide_temp,
    int step_mode, int noise_mode, const float* __restrict__ noise, uint64_t seed, uint64_t offset,
    uint64_t row_offset, int64_t* __restrict__ x_out, float* __restrict__ probs_out, int* __restrict__ nan_flag) {
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 5;
  const int k = threadIdx.x & 31;
  if (row >= rows) return;
  const bool act = k < K;
  const int xt = (int)x[row];
  float z = act ? logits[(size_t)row * K + k] : -INFINITY;
  const float zmax = half_max(z);
  const float se = half_sum(act ? expf(z - zmax) : 0.f);
  const float lp = z - (zmax + logf(se));
  const float px = act ? expf(lp) : 0.f;  // x_theta
  const float alpha_s = 1.f - ms, alpha_t = 1.f - mt;
  const float alpha_ts = alpha_t / alpha_s;
  const float d_alpha = alpha_s - alpha_t;
  const float oh = (k == xt) ? 1.f : 0.f;
  const float x_at_xt = __shfl(px, (threadIdx.x & 32) + xt);  // gather(x, -1, xt)
  const float lim = 1.0f / (float)K;
  const float num = alpha_t * (float)K * px * oh + (alpha_ts - alpha_t) * oh + d_alpha * px +
                    (1.f - alpha_ts) * (1.f - alpha_s) * lim;
  const float den = alpha_t * (float)K * x_at_xt + (1.f - alpha_t);
  float q = act ? num / den : 0.f;
  const size_t lidx = (size_t)row * K + k;
  const uint64_t gidx = ((uint64_t)row + row_offset) * (uint64_t)K + k;
  if (step_mode != SGPO_STEP_SOFT_CG) {
    if (probs_out && act) probs_out[lidx] = q;
    if (step_mode == SGPO_STEP_SOFT) return;
    float dn = act ? noise_denominator(noise_mode, noise, lidx, seed, offset, gidx, false) : 1.f;
    float v = act ? q / dn : -INFINITY;
    int samp = half_argmax(v, k);
    if (k == 0) x_out[row] = samp;
    return;
  }
  // classifier guidance (cls_guidance.py:38-52, models/mdlm.py:399-411), same operation order as the MDLM kernel
  const bool guided = log_ratio != nullptr && guide_temp != 0.f;
  if (k == xt) q -= 1.0f;
  if (guided && act) {
    const int l = row % L;
    const int slot = slot_of_pos ? slot_of_pos[l] : l;
    if (slot >= 0) {
      float lr = log_ratio[((size_t)(row / L) * n_slots + slot) * K + k] / guide_temp;
      if (lr == lr) lr = fminf(lr, 80.0f);  // torch.clamp propagates NaN
      q = q * expf(lr);
    }
    if (nan_flag && q != q) atomicOr(nan_flag, 1);
  }
  if (k == xt) q += 1.0f;
  q = q < 0.f ? 0.f : q;   // clamp(min=0) keeps a NaN
  if (!act) q = 0.f;
  const float tot = half_sum(q);
  if (nan_flag && k == 0 && !(tot > 0.f && tot <= 3.4028234e38f)) atomicOr(nan_flag, 2);
  const float p = q / tot;
  if (probs_out && act) probs_out[lidx] = p;
  float dn = act ? noise_denominator(noise_mode, noise, lidx, seed, offset, gidx, true) : 1.f;
  float v = act ? p / dn : -INFINITY;
  int samp = half_argmax(v, k);
  if (k == 0) x_out[row] = samp;
}

static int launch_reverse_step_udlm(const float* logits, const int64_t* x, const float* log_ratio, const int* slot_of_pos,
                                    int n_slots, int B, int L, int K, float move_chance_t, float move_chance_s,
                                    float guide_temp, int step_mode, int noise_mode, const float* noise, uint64_t seed,
                                    uint64_t offset, uint64_t row_offset, int64_t* x_out, float* probs_out, int* nan_flag,
                                    void* stream) {
  SGPO_REQUIRE(logits && x, "reverse_step_udlm: null logits/x");
  SGPO_REQUIRE(B > 0 && L > 0 && K > 0 && K <= 32, "reverse_step_udlm: B=%d L=%d K=%d (K <= 32)", B, L, K);
  SGPO_REQUIRE(step_mode >= 0 && step_mode <= 2, "reverse_step_udlm: bad step_mode %d", step_mode);
  SGPO_REQUIRE(noise_mode >= 0 && noise_mode <= 3, "reverse_step_udlm: bad noise_mode %d", noise_mode);
  SGPO_REQUIRE(step_mode == SGPO_STEP_SOFT ? probs_out != nullptr : x_out != nullptr, "reverse_step_udlm: output missing");
  SGPO_REQUIRE(noise_mode == SGPO_NOISE_PHILOX || noise_mode == SGPO_NOISE_NONE || step_mode == SGPO_STEP_SOFT || noise, "reverse_step_udlm: noise tensor missing");
  SGPO_REQUIRE(log_ratio == nullptr || slot_of_pos != nullptr || n_slots == L, "reverse_step_udlm: n_slots must equal L without a slot map");
  const int rows = B * L;
  ProfScope prof_scope(PK_REVERSE_STEP, (hipStream_t)stream,
                       (double)rows * (K * 4.0 * ((log_ratio && guide_temp != 0.f && step_mode == SGPO_STEP_SOFT_CG) ? 2 : 1) + 16.0));
  hipLaunchKernelGGL(reverse_step_udlm_kernel, dim3((rows + 7) / 8), dim3(256), 0, (hipStream_t)stream, logits, x, log_ratio,
                     slot_of_pos, n_slots, rows, L, K, move_chance_t, move_chance_s, guide_temp, step_mode, noise_mode, noise,
                     seed, offset, row_offset, x_out, probs_out, nan_flag);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

extern "C" int sgpo_reverse_step_udlm(const float* logits, const int64_t* x, int B, int L, int K,
                                      float move_chance_t, float move_chance_s, int step_mode, int noise_mode,
                                      const float* noise, uint64_t seed, uint64_t offset, uint64_t row_offset,
                                      int64_t* x_out, float* probs_out, void* stream) {
  SGPO_REQUIRE(step_mode == SGPO_STEP_SOFT || step_mode == SGPO_STEP_HARD, "reverse_step_udlm: step_mode must be SOFT or HARD");
  return launch_reverse_step_udlm(logits, x, nullptr, nullptr, L, B, L, K, move_chance_t, move_chance_s, 0.f, step_mode,
                                  noise_mode, noise, seed, offset, row_offset, x_out, probs_out, nullptr, stream);
}

extern "C" int sgpo_guided_step_udlm(const float* logits, const int64_t* x, const float* log_ratio,
                                     const int* slot_of_pos, int n_slots, int B, int L, int K, float move_chance_t,
                                     float move_chance_s, float guide_temp, int noise_mode, const float* noise,
                                     uint64_t seed, uint64_t offset, uint64_t row_offset, int64_t* x_out,
                                     float* probs_out, int* nan_flag, void* stream) {
  return launch_reverse_step_udlm(logits, x, log_ratio, slot_of_pos, n_slots, B, L, K, move_chance_t, move_chance_s,
                                  guide_temp, SGPO_STEP_SOFT_CG, noise_mode, noise, seed, offset, row_offset, x_out,
                                  probs_out, nan_flag, stream);
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void q_sample_mdlm_kernel(const int64_t* __restrict__ x, const float* __restrict__ mc,
                                                            int rows, int L, int mask_index, int noise_mode,
                                                            const float* __restrict__ noise, uint64_t seed, uint64_t offset,
                                                            uint64_t row_offset, int64_t* __restrict__ x_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows) return;
  float u;
  if (noise_mode == SGPO_NOISE_PHILOX) {
    Philox4 r = philox4x32_10(seed, (uint64_t)i + row_offset, offset);
    u = (float)(r.x >> 8) * (1.0f / 16777216.0f);  // [0,1) like torch.rand
  } else {
    u = noise[i];
  }
  x_out[i] = (u < mc[i / L]) ? (int64_t)mask_index : x[i];
}

extern "C" int sgpo_q_sample_mdlm(const int64_t* x, const float* move_chance, int B, int L, int mask_index,
                                  int noise_mode, const float* noise, uint64_t seed, uint64_t offset,
                                  uint64_t row_offset, int64_t* x_out, void* stream) {
  SGPO_REQUIRE(x && move_chance && x_out, "q_sample_mdlm: null argument");
  SGPO_REQUIRE(B > 0 && L > 0, "q_sample_mdlm: B=%d L=%d", B, L);
  SGPO_REQUIRE(noise_mode == SGPO_NOISE_PHILOX || (noise_mode == SGPO_NOISE_UNIFORM && noise), "q_sample_mdlm: noise_mode must be 0 or 2 (+tensor)");
  const int rows = B * L;
  hipLaunchKernelGGL(q_sample_mdlm_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, move_chance,
                     rows, L, mask_index, noise_mode, noise, seed, offset, row_offset, x_out);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

// ---------------------------------------------------------------------------
// q_xt, uniform branch (mdlm_diffusion.py:593-601): move = rand < move_chance; token = randint(0, vocab); where(move, token, x).
__global__ __launch_bounds__(256) void q_sample_udlm_kernel(const int64_t* __restrict__ x, const float* __restrict__ mc,
                                                            int rows, int L, int vocab, int noise_mode,
                                                            const float* __restrict__ noise, const int64_t* __restrict__ noise_tokens,
                                                            uint64_t seed, uint64_t offset, uint64_t row_offset,
                                                            int64_t* __restrict__ x_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows) return;
  float u;
  int64_t tok;
  if (noise_mode == SGPO_NOISE_PHILOX) {
    Philox4 r = philox4x32_10(seed, (uint64_t)i + row_offset, offset);
    u = (float)(r.x >> 8) * (1.0f / 16777216.0f);  // [0,1) like torch.rand
    tok = (int64_t)(((uint64_t)r.y * (uint64_t)vocab) >> 32);  // uniform over [0, vocab)
  } else {
    u = noise[i];
    tok = noise_tokens[i];
  }
  x_out[i] = (u < mc[i / L]) ? tok : x[i];
}

extern "C" int sgpo_q_sample_udlm(const int64_t* x, const float* move_chance, int B, int L, int vocab, int noise_mode,
                                  const float* noise, const int64_t* noise_tokens, uint64_t seed, uint64_t offset,
                                  uint64_t row_offset, int64_t* x_out, void* stream) {
  SGPO_REQUIRE(x && move_chance && x_out, "q_sample_udlm: null argument");
  SGPO_REQUIRE(B > 0 && L > 0 && vocab > 0, "q_sample_udlm: B=%d L=%d vocab=%d", B, L, vocab);
  SGPO_REQUIRE(noise_mode == SGPO_NOISE_PHILOX || (noise_mode == SGPO_NOISE_UNIFORM && noise && noise_tokens),
               "q_sample_udlm: noise_mode must be 0 or 2 (+ uniforms and tokens)");
  const int rows = B * L;
  hipLaunchKernelGGL(q_sample_udlm_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, move_chance,
                     rows, L, vocab, noise_mode, noise, noise_tokens, seed, offset, row_offset, x_out);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void categorical_kernel(const float* __restrict__ probs, int64_t rows, int K,
                                                          int noise_mode, const float* __restrict__ noise, uint64_t seed,
                                                          uint64_t offset, uint64_t row_offset, int64_t* __restrict__ out) {
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
  const int k = threadIdx.x & 31;
  if (row >= rows) return;
  const bool act = k < K;
  const size_t lidx = (size_t)row * K + k;
  const float p = act ? probs[lidx] : 0.f;
  const uint64_t gidx = ((uint64_t)row + row_offset) * (uint64_t)K + k;
  float den = act ? noise_denominator(noise_mode, noise, lidx, seed, offset, gidx, noise_mode != SGPO_NOISE_UNIFORM) : 1.f;
  float v = act ? p / den : -INFINITY;
  int samp = half_argmax(v, k);
  if (k == 0) out[row] = samp;
}

extern "C" int sgpo_categorical_sample(const float* probs, int64_t rows, int K, int noise_mode, const float* noise,
                                       uint64_t seed, uint64_t offset, uint64_t row_offset, int64_t* out, void* stream) {
  SGPO_REQUIRE(probs && out, "categorical_sample: null argument");
  SGPO_REQUIRE(rows > 0 && K > 0 && K <= 32, "categorical_sample: rows=%lld K=%d (K <= 32)", (long long)rows, K);
  SGPO_REQUIRE(noise_mode == SGPO_NOISE_PHILOX || noise_mode == SGPO_NOISE_NONE || noise, "categorical_sample: noise tensor missing");
  hipLaunchKernelGGL(categorical_kernel, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, probs, rows,
                     K, noise_mode, noise, seed, offset, row_offset, out);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}
