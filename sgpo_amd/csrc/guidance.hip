// Value-function kernels: MLPModel forward and the exact-guidance jump log-ratios in
// factored (rank-1 update) form.
//
// Reference: problem/protein.py:58-121 (MLPModel), models/mdlm.py:111-162,360-375 and
// models/d3pm.py:55-106,146-161 (all single-site jumps, materialised as B*D*S x D there).
//
// A jump changes one one-hot row, so the first Linear's pre-activation moves by a table
// difference:  pre(x with x_d <- s) = pre(x) + T1[d][s] - T1[d][x_d],
//   T1[i][s][:] = W1[:, i*h:(i+1)*h] @ E[:, s]      (E = embedder.weight, bias folded into be1)
// The table (n_read x alphabet x 64 fp32, <= 944 KB) is L2-resident.  The 64x64 tails run on
// the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fma chain), one 32-lane
// tile = the 31(+1) jumps of one (sequence, read position), "transposed" so that the
// activation accumulator of one layer is the B operand of the next with no data movement.
#include <mutex>
#include <vector>

#include "../../include/sgpo_hip.h"
#include "common.h"
#include "prof.h"
#include "mlp.h"

// ---- setup kernels ---------------------------------------------------------
__global__ void mlp_table_kernel(const float* __restrict__ W1, int in_dim, const float* __restrict__ E /*[64][A]*/,
                                 int A, int n_read, float* __restrict__ T1) {
  // T1[i][s][j] = sum_k W1[j][i*64+k] * E[k][s]
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_read * A * MH) return;
  const int j = idx % MH, s = (idx / MH) % A, i = idx / (MH * A);
  const float* w = W1 + (size_t)j * in_dim + i * MH;
  float a = 0.f;
  for (int k = 0; k < MH; ++k) a = fmaf(w[k], E[k * A + s], a);
  T1[idx] = a;
}
__global__ void mlp_bias_kernel(const float* __restrict__ W1, int in_dim, const float* __restrict__ be,
                                const float* __restrict__ b1, int n_read, float* __restrict__ be1) {
  const int j = threadIdx.x;
  if (j >= MH) return;
  const float* w = W1 + (size_t)j * in_dim;
  float a = 0.f;
  for (int k = 0; k < n_read * MH; ++k) a = fmaf(w[k], be[k % MH], a);
  be1[j] = b1[j] + a;
}
__global__ void transpose_kernel(const float* __restrict__ src, int ld, int col0, float* __restrict__ dst) {
  // dst[k][j] = src[j][col0 + k], 64x64
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= MH * MH) return;
  const int k = idx / MH, j = idx % MH;
  dst[idx] = src[(size_t)j * ld + col0 + k];
}
__device__ __host__ inline int acc_unit(int t, int r, int h) { return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h; }
__global__ void permute_layer_kernel(const float* __restrict__ W /*[64][64] row-major [out][in]*/, float* __restrict__ dst) {
  // dst[to][ti][r][h][i] = W[32 to + i][unit(ti, r, h)]
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= MH * MH) return;
  const int i = idx & 31, h = (idx >> 5) & 1, r = (idx >> 6) & 15, ti = (idx >> 10) & 1, to = (idx >> 11) & 1;
  dst[idx] = W[(32 * to + i) * MH + acc_unit(ti, r, h)];
}
__global__ void permute_bias_kernel(const float* __restrict__ b, float* __restrict__ dst) {
  // dst[t][r][h] = b[unit(t, r, h)]
  const int idx = threadIdx.x;
  if (idx >= MH) return;
  const int h = idx & 1, r = (idx >> 1) & 15, t = idx >> 5;
  dst[idx] = b[acc_unit(t, r, h)];
}

extern "C" void sgpo_mlp_destroy(sgpo_mlp* m) {
  if (!m) return;
  (void)hipFree(m->read_pos); (void)hipFree(m->params); (void)hipFree(m->T1); (void)hipFree(m->be1);
  (void)hipFree(m->w1t_time); (void)hipFree(m->wt_lin_t);
  for (int i = 0; i < 3; ++i) (void)hipFree(m->w_t[i]);
  (void)hipFree(m->w_perm); (void)hipFree(m->bias_perm);
  delete[] m->read_pos_host;
  delete m;
}

// dst <- the weights of src (same architecture and read positions), asynchronously on `stream`: every device table is
// overwritten in place, so launches recorded with dst's pointers (a HIP graph of stacked Thompson draws) see the new
// member on their next replay.
extern "C" int sgpo_mlp_copy(sgpo_mlp* dst, const sgpo_mlp* src, void* stream) {
  SGPO_REQUIRE(dst && src, "mlp_copy: null argument");
  SGPO_REQUIRE(dst->n_read == src->n_read && dst->alphabet == src->alphabet && dst->hidden == src->hidden &&
               dst->time_cond == src->time_cond && dst->n_params == src->n_params,
               "mlp_copy: the value functions differ in shape (%d/%d read positions, %d/%d letters)", dst->n_read, src->n_read,
               dst->alphabet, src->alphabet);
  for (int i = 0; i < src->n_read; ++i)
    SGPO_REQUIRE(dst->read_pos_host[i] == src->read_pos_host[i], "mlp_copy: read position %d differs", i);
  if (dst == src) return SGPO_OK;
  hipStream_t st = (hipStream_t)stream;
  const size_t h = MH, A = src->alphabet;
#define CP(field, n) SGPO_CHECK_HIP(hipMemcpyAsync(dst->field, src->field, (n) * sizeof(float), hipMemcpyDeviceToDevice, st))
  CP(params, src->n_params);
  CP(T1, (size_t)src->n_read * A * h);
  CP(be1, h);
  CP(w1t_time, h * h);
  CP(wt_lin_t, h * h);
  for (int i = 0; i < 3; ++i) CP(w_t[i], h * h);
  CP(w_perm, 3 * h * h);
  CP(bias_perm, 4 * h);
#undef CP
  return SGPO_OK;
}

extern "C" int sgpo_mlp_create(const float* blob, size_t n_floats, int n_read, const int* read_pos, int alphabet,
                               int hidden, int time_conditioned, sgpo_mlp** out) {
  SGPO_REQUIRE(blob && read_pos && out, "mlp_create: null argument");
  SGPO_REQUIRE(hidden == MH, "mlp_create: hidden_dim %d unsupported (64 only)", hidden);
  SGPO_REQUIRE(n_read > 0 && alphabet > 0 && alphabet <= 32, "mlp_create: n_read=%d alphabet=%d (alphabet <= 32)", n_read, alphabet);
  const size_t h = MH, A = alphabet;
  const size_t in_dim = h * (n_read + (time_conditioned ? 1 : 0));
  const size_t expect = h / 2 + h * h + h + h * A + h + h * in_dim + h + 3 * (h * h + h) + h + 1;
  SGPO_REQUIRE(n_floats == expect, "mlp_create: blob has %zu floats, expected %zu", n_floats, expect);
  sgpo_mlp* m = new sgpo_mlp();
  m->n_read = n_read; m->alphabet = alphabet; m->hidden = hidden; m->time_cond = time_conditioned;
#define HIPF(e) do { hipError_t _e = (e); if (_e != hipSuccess) { sgpo_set_error("%s: %s", #e, hipGetErrorString(_e)); sgpo_mlp_destroy(m); return SGPO_ERR_HIP; } } while (0)
  m->n_params = n_floats;
  HIPF(hipMalloc((void**)&m->params, n_floats * 4));
  HIPF(hipMemcpy(m->params, blob, n_floats * 4, hipMemcpyDeviceToDevice));
  HIPF(hipMalloc((void**)&m->read_pos, n_read * sizeof(int)));
  HIPF(hipMemcpy(m->read_pos, read_pos, n_read * sizeof(int), hipMemcpyHostToDevice));
  m->read_pos_host = new int[n_read];
  for (int i = 0; i < n_read; ++i) m->read_pos_host[i] = read_pos[i];
  HIPF(hipMalloc((void**)&m->T1, (size_t)n_read * A * h * 4));
  HIPF(hipMalloc((void**)&m->be1, h * 4));
  HIPF(hipMalloc((void**)&m->w1t_time, h * h * 4));
  HIPF(hipMalloc((void**)&m->wt_lin_t, h * h * 4));
  for (int i = 0; i < 3; ++i) HIPF(hipMalloc((void**)&m->w_t[i], h * h * 4));
  HIPF(hipMalloc((void**)&m->w_perm, 3 * h * h * 4));
  HIPF(hipMalloc((void**)&m->bias_perm, 4 * h * 4));
  const float* p = m->params;
  auto take = [&](size_t n) { const float* r = p; p += n; return r; };
  m->tW = take(h / 2);
  const float* wt_lin = take(h * h); m->bt = take(h);
  const float* E = take(h * A); const float* be = take(h);
  const float* W1 = take(h * in_dim); const float* b1 = take(h);
  const float* W2 = take(h * h); m->b2 = take(h);
  const float* W3 = take(h * h); m->b3 = take(h);
  const float* W4 = take(h * h); m->b4 = take(h);
  m->W2 = W2; m->W3 = W3; m->W4 = W4;
  m->w5 = take(h); m->b5 = take(1);
  const int nT = n_read * (int)A * MH;
  hipLaunchKernelGGL(mlp_table_kernel, dim3((nT + 255) / 256), dim3(256), 0, 0, W1, (int)in_dim, E, (int)A, n_read, m->T1);
  hipLaunchKernelGGL(mlp_bias_kernel, dim3(1), dim3(64), 0, 0, W1, (int)in_dim, be, b1, n_read, m->be1);
  if (time_conditioned)
    hipLaunchKernelGGL(transpose_kernel, dim3(16), dim3(256), 0, 0, W1, (int)in_dim, n_read * MH, m->w1t_time);
  hipLaunchKernelGGL(transpose_kernel, dim3(16), dim3(256), 0, 0, wt_lin, MH, 0, m->wt_lin_t);
  const float* Ws[3] = {W2, W3, W4};
  const float* bs[4] = {m->b2, m->b3, m->b4, m->w5};
  for (int i = 0; i < 3; ++i) {
    hipLaunchKernelGGL(transpose_kernel, dim3(16), dim3(256), 0, 0, Ws[i], MH, 0, m->w_t[i]);
    hipLaunchKernelGGL(permute_layer_kernel, dim3(16), dim3(256), 0, 0, Ws[i], m->w_perm + (size_t)i * h * h);
  }
  for (int i = 0; i < 4; ++i)
    hipLaunchKernelGGL(permute_bias_kernel, dim3(1), dim3(64), 0, 0, bs[i], m->bias_perm + i * MH);
  HIPF(hipGetLastError());
  HIPF(hipDeviceSynchronize());
#undef HIPF
  *out = m;
  return SGPO_OK;
}

// ---- forward of one sequence per wave (VALU; also yields the layer-1 pre-activation) -------
struct MlpDev {
  const int* read_pos; const float* T1; const float* be1; const float* w1t_time; const float* wt_lin_t;
  const float* w2t; const float* w3t; const float* w4t;
  const float *tW, *bt, *b2, *b3, *b4, *w5, *b5;
  int n_read, alphabet, time_cond;
};

// lists (absorbing-state compaction, mask_index >= 0): per sequence cnt[2] = {#masked, #unmasked read slots},
// then the masked slots and the unmasked slots, each in increasing order (uint16 [2][n_read])
__device__ __forceinline__ void mlp_base_body(const MlpDev& d, int b, const int64_t* __restrict__ x, const float* __restrict__ t,
                                              float t_scalar, int L, float* __restrict__ hpre,
                                              float* __restrict__ fout, int mask_index, int* __restrict__ cnt,
                                              unsigned short* __restrict__ lists) {
  __shared__ float buf[2][MH];
  const int j = threadIdx.x;
  float pre = d.be1[j];
  if (d.time_cond) {
    const float tv = t ? t[b] : t_scalar;
    // GaussianFourierProjection (problem/protein.py:54-56): x*W*2*pi, cat(sin, cos)
    const float arg = tv * d.tW[j & 31] * 2.0f * 3.14159274101257324f;
    buf[0][j] = (j < 32) ? sinf(arg) : cosf(arg);
    __syncthreads();
    float te = d.bt[j];
    for (int k = 0; k < MH; ++k) te = fmaf(d.wt_lin_t[k * MH + j], buf[0][k], te);
    buf[1][j] = te;
    __syncthreads();
    for (int k = 0; k < MH; ++k) pre = fmaf(d.w1t_time[k * MH + j], buf[1][k], pre);
    __syncthreads();
  }
  const int64_t* xr = x + (size_t)b * L;
  // eight table rows in flight per lane (one sequence is one wave: a dependent load per residue is pure latency);
  // the sum keeps its order
  for (int i0 = 0; i0 < d.n_read; i0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u;
      v[u] = 0.f;
      if (i < d.n_read) {
        int tok = (int)xr[d.read_pos[i]];
        tok = tok < 0 ? 0 : (tok >= d.alphabet ? d.alphabet - 1 : tok);
        v[u] = d.T1[((size_t)i * d.alphabet + tok) * MH + j];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u < d.n_read) pre += v[u];
  }
  if (hpre) hpre[(size_t)b * MH + j] = pre;
  if (cnt) {
    unsigned short* ml = lists + (size_t)b * 2 * d.n_read;
    unsigned short* ul = ml + d.n_read;
    int nm = 0, nu = 0;
    for (int i0 = 0; i0 < d.n_read; i0 += 64) {
      const int i = i0 + j;
      const bool in = i < d.n_read;
      const bool masked = in && (int)xr[d.read_pos[in ? i : 0]] == mask_index;
      const unsigned long long bm = __ballot(masked), bu = __ballot(in && !masked);
      const unsigned long long below = (1ull << j) - 1ull;
      if (masked) ml[nm + __popcll(bm & below)] = (unsigned short)i;
      if (in && !masked) ul[nu + __popcll(bu & below)] = (unsigned short)i;
      nm += __popcll(bm); nu += __popcll(bu);
    }
    if (j == 0) { cnt[2 * b] = nm; cnt[2 * b + 1] = nu; }
  }
  if (!fout) return;
  buf[0][j] = fmaxf(pre, 0.f);
  __syncthreads();
  float a = d.b2[j];
  for (int k = 0; k < MH; ++k) a = fmaf(d.w2t[k * MH + j], buf[0][k], a);
  buf[1][j] = fmaxf(a, 0.f);
  __syncthreads();
  a = d.b3[j];
  for (int k = 0; k < MH; ++k) a = fmaf(d.w3t[k * MH + j], buf[1][k], a);
  buf[0][j] = a;
  __syncthreads();
  a = d.b4[j];
  for (int k = 0; k < MH; ++k) a = fmaf(d.w4t[k * MH + j], buf[0][k], a);
  a = fmaxf(a, 0.f) * d.w5[j];
  a = wave_sum(a);
  if (j == 0) fout[b] = a + d.b5[0];
}

__global__ __launch_bounds__(64) void mlp_base_kernel(MlpDev d, const int64_t* __restrict__ x, const float* __restrict__ t,
                                                      float t_scalar, int L, float* __restrict__ hpre,
                                                      float* __restrict__ fout, int mask_index, int* __restrict__ cnt,
                                                      unsigned short* __restrict__ lists) {
  mlp_base_body(d, blockIdx.x, x, t, t_scalar, L, hpre, fout, mask_index, cnt, lists);
}

// several stacked draws (mlp.h): blockIdx.y picks the draw and its value function, blockIdx.x a sequence of it; the
// shared buffers are indexed by the row of the stacked batch
struct MlpGroups {
  int seq_lo[SGPO_MAX_GROUPS], n_seq[SGPO_MAX_GROUPS];
  MlpDev d[SGPO_MAX_GROUPS];
};
__global__ __launch_bounds__(64) void mlp_base_groups_kernel(MlpGroups gs, const int64_t* __restrict__ x, float t_scalar, int L,
                                                             float* __restrict__ hpre, int mask_index, int* __restrict__ cnt,
                                                             unsigned short* __restrict__ lists) {
  const int g = blockIdx.y;
  if ((int)blockIdx.x >= gs.n_seq[g]) return;
  mlp_base_body(gs.d[g], gs.seq_lo[g] + blockIdx.x, x, nullptr, t_scalar, L, hpre, nullptr, mask_index, cnt, lists);
}

static MlpDev mlp_dev(const sgpo_mlp* m) {
  MlpDev d;
  d.read_pos = m->read_pos; d.T1 = m->T1; d.be1 = m->be1; d.w1t_time = m->w1t_time; d.wt_lin_t = m->wt_lin_t;
  d.w2t = m->w_t[0]; d.w3t = m->w_t[1]; d.w4t = m->w_t[2];
  d.tW = m->tW; d.bt = m->bt; d.b2 = m->b2; d.b3 = m->b3; d.b4 = m->b4; d.w5 = m->w5; d.b5 = m->b5;
  d.n_read = m->n_read; d.alphabet = m->alphabet; d.time_cond = m->time_cond;
  return d;
}

// every read position must lie inside a row of length L (checked on the host copy; also for negative entries)
static int check_read_pos(const sgpo_mlp* m, int L, const char* who) {
  for (int i = 0; i < m->n_read; ++i)
    SGPO_REQUIRE(m->read_pos_host[i] >= 0 && m->read_pos_host[i] < L, "%s: value function reads position %d of a length-%d row", who,
                 m->read_pos_host[i], L);
  return SGPO_OK;
}

extern "C" int sgpo_mlp_forward(sgpo_mlp* m, const int64_t* x, const float* t, int B, int L, float* out, void* stream) {
  SGPO_REQUIRE(m && x && out, "mlp_forward: null argument");
  SGPO_REQUIRE(B > 0 && L > 0, "mlp_forward: B=%d L=%d", B, L);
  if (int rc = check_read_pos(m, L, "mlp_forward")) return rc;
  hipLaunchKernelGGL(mlp_base_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, mlp_dev(m), x, t, 0.f, L,
                     (float*)nullptr, out, -1, (int*)nullptr, (unsigned short*)nullptr);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

// ---- jump kernel --------------------------------------------------------------------
// One wave-tile = the S (<= 32) jumps of one (sequence b, read slot i); lane column = target
// state s, both lane halves hold complementary halves of the 64 hidden units in accumulator
// order unit(t, r, h) = 32 t + (r&3) + 8 (r>>2) + 4 h.
__device__ __forceinline__ void mfma_layer(const float* __restrict__ wl /*LDS [to][ti][r][h][i]*/, const f32x16 (&in)[2],
                                           f32x16 (&out)[2], int lane) {
#pragma unroll
  for (int to = 0; to < 2; ++to) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float a = wl[((to * 2 + ti) * 16 + r) * 64 + lane];  // [h][i] == lane order
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, in[ti][r], acc, 0, 0, 0);
      }
    out[to] = acc;
  }
}

// Two tile kinds.  FULL: the S targets of one (b, slot).  PACKED (absorbing-state diffusion only): up to 31
// unmasked slots of one sequence, each with the single target `mask_index`, plus the identity jump in
// column 31 (f(x) with the same summation order, so the ratios equal the FULL tile's bit for bit).  At a
// position whose token is not the mask, the prior rate is exactly 0 for every target except the mask and
// the token itself (models/mdlm.py subs parameterisation + carry-over), so those are the only jumps that
// can change the guided rates; the rest of the row stays at the zero the caller's memset left there.
__device__ __forceinline__ void mlp_jump_body(const float* __restrict__ w_perm, const float* __restrict__ bias_perm,
                                              const float* __restrict__ T1, const int* __restrict__ read_pos,
                                              const float* __restrict__ hpre, const int64_t* __restrict__ x,
                                              float b5, int B, int n_read, int alphabet, int S, int L,
                                              int mask_index, const int* __restrict__ cnt,
                                              const unsigned short* __restrict__ lists, int spread,
                                              float* __restrict__ log_ratio) {
  __shared__ float wl[3 * MH * MH];
  __shared__ float bl[4 * MH];
  for (int i = threadIdx.x; i < 3 * MH * MH; i += 256) wl[i] = w_perm[i];
  for (int i = threadIdx.x; i < 4 * MH; i += 256) bl[i] = bias_perm[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & 31, h = lane >> 5;
  const int n_waves = gridDim.x * 4;
  const bool compact = cnt != nullptr;
  const int mask_tok = mask_index < 0 ? 0 : (mask_index >= alphabet ? alphabet - 1 : mask_index);
  // compact: a wave walks whole sequences (their tile counts differ) -- or, for small batches (`spread`), one
  // tile slot of a sequence, so that a handful of sequences still spreads over the GPU; otherwise tiles are
  // dealt singly
  const int per = (compact && !spread) ? 1 : n_read;
  const int n_units = B * per;
  for (int unit = blockIdx.x * 4 + wave; unit < n_units; unit += n_waves) {
    const int b = unit / per;
    int nm = 1, nu = 0;
    if (compact) { nm = cnt[2 * b]; nu = cnt[2 * b + 1]; }
    const int n_t = compact ? nm + (nu + 30) / 31 : 1;
    const unsigned short* ml = compact ? lists + (size_t)b * 2 * n_read : nullptr;
    const float* hp = hpre + (size_t)b * MH;
    const int j0 = (compact && spread) ? unit % per : 0;
    const int j1 = (compact && spread) ? (j0 + 1 < n_t ? j0 + 1 : n_t) : n_t;
    for (int j = j0; j < j1; ++j) {
      // The 192 weight fragments are the same for every tile; hoisted out of this loop they would need
      // 192 VGPRs and spill to scratch.  An opaque (always zero) offset keeps the LDS reads inside.
      int wofs = 0;
      asm volatile("" : "+s"(wofs));
      const float* wt = wl + wofs;
      const bool packed = compact && j >= nm;
      // this lane's jump: slot i, target token sc; `ident`: no jump at all (f(x))
      int i, sc;
      bool live = true, ident = false;
      if (!packed) {
        i = compact ? (int)ml[j] : unit % n_read;
        sc = s < S ? s : S - 1;
        live = s < S;
      } else {
        const int u = (j - nm) * 31 + s;
        ident = (s == 31);
        live = !ident && u < nu;
        i = live ? (int)ml[n_read + u] : (int)ml[n_read];   // any valid slot for idle lanes
        sc = mask_tok;
      }
      int xi = (int)x[(size_t)b * L + read_pos[i]];
      xi = xi < 0 ? 0 : (xi >= alphabet ? alphabet - 1 : xi);
      const float* tn = T1 + ((size_t)i * alphabet + sc) * MH;
      const float* to_ = T1 + ((size_t)i * alphabet + xi) * MH;
      const bool same = ident || (sc == xi);  // identity jump: exactly f(x)
      f32x16 a[2], o[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int u = 32 * t + 8 * g + 4 * h;
          const float4 p4 = *reinterpret_cast<const float4*>(hp + u);
          const float4 n4 = *reinterpret_cast<const float4*>(tn + u);
          const float4 o4 = *reinterpret_cast<const float4*>(to_ + u);
          a[t][g * 4 + 0] = fmaxf(same ? p4.x : (p4.x + n4.x - o4.x), 0.f);
          a[t][g * 4 + 1] = fmaxf(same ? p4.y : (p4.y + n4.y - o4.y), 0.f);
          a[t][g * 4 + 2] = fmaxf(same ? p4.z : (p4.z + n4.z - o4.z), 0.f);
          a[t][g * 4 + 3] = fmaxf(same ? p4.w : (p4.w + n4.w - o4.w), 0.f);
        }
      // mlp.2 + ReLU
      mfma_layer(wt, a, o, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = fmaxf(o[t][r] + bl[0 * MH + (t * 16 + r) * 2 + h], 0.f);
      // mlp.4 (no activation)
      mfma_layer(wt + MH * MH, o, a, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[t][r] = a[t][r] + bl[1 * MH + (t * 16 + r) * 2 + h];
      // cls_head.0 + ReLU
      mfma_layer(wt + 2 * MH * MH, a, o, lane);
      float f = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = fmaxf(o[t][r] + bl[2 * MH + (t * 16 + r) * 2 + h], 0.f);
          f = fmaf(v, bl[3 * MH + (t * 16 + r) * 2 + h], f);
        }
      f += __shfl_xor(f, 32);
      f += b5;
      const float fbase = __shfl(f, packed ? 31 : xi);   // FULL: xi is the same in every lane
      if (h == 0 && live) log_ratio[((size_t)b * n_read + i) * S + sc] = (sc == xi) ? 0.f : f - fbase;
    }
  }
}

__global__ __launch_bounds__(256, 2) void mlp_jump_kernel(const float* __restrict__ w_perm, const float* __restrict__ bias_perm,
                                                       const float* __restrict__ T1, const int* __restrict__ read_pos,
                                                       const float* __restrict__ hpre, const int64_t* __restrict__ x,
                                                       float b5, int B, int n_read, int alphabet, int S, int L,
                                                       int mask_index, const int* __restrict__ cnt,
                                                       const unsigned short* __restrict__ lists, int spread,
                                                       float* __restrict__ log_ratio) {
  mlp_jump_body(w_perm, bias_perm, T1, read_pos, hpre, x, b5, B, n_read, alphabet, S, L, mask_index, cnt, lists, spread, log_ratio);
}

// several stacked draws: blockIdx.y picks the draw; its rows of the shared buffers are addressed by offsetting the
// pointers, so inside the draw every index is the one of its own launch
struct JumpGroups {
  int seq_lo[SGPO_MAX_GROUPS], n_seq[SGPO_MAX_GROUPS];
  const float* w_perm[SGPO_MAX_GROUPS];
  const float* bias_perm[SGPO_MAX_GROUPS];
  const float* T1[SGPO_MAX_GROUPS];
};
__global__ __launch_bounds__(256, 2) void mlp_jump_groups_kernel(JumpGroups gs, const int* __restrict__ read_pos,
                                                              const float* __restrict__ hpre, const int64_t* __restrict__ x,
                                                              int n_read, int alphabet, int S, int L, int mask_index,
                                                              const int* __restrict__ cnt, const unsigned short* __restrict__ lists,
                                                              float* __restrict__ log_ratio) {
  const int g = blockIdx.y;
  const int lo = gs.seq_lo[g], n = gs.n_seq[g];
  if (n <= 0) return;
  mlp_jump_body(gs.w_perm[g], gs.bias_perm[g], gs.T1[g], read_pos, hpre + (size_t)lo * MH, x + (size_t)lo * L, 0.f, n, n_read,
                alphabet, S, L, mask_index, cnt + 2 * lo, lists + (size_t)lo * 2 * n_read, n < 1024 ? 1 : 0,
                log_ratio + (size_t)lo * n_read * S);
}

static size_t al256(size_t n) { return (n + 255) & ~(size_t)255; }

// hpre [B][64] f32 | cnt [B][2] i32 | lists [B][2][n_read] u16
extern "C" size_t sgpo_mlp_workspace_bytes(const sgpo_mlp* m, int B) {
  if (!m || B <= 0) return 0;
  return al256((size_t)B * MH * sizeof(float)) + al256((size_t)B * 2 * sizeof(int)) +
         al256((size_t)B * 2 * m->n_read * sizeof(unsigned short));
}

// log_ratio <- 0 by a kernel of the library's own: a hipMemsetAsync recorded into a HIP graph as a memset node did not
// reliably clear the buffer on replay (round 3: NaN rates at a later step of a recorded sampling loop)
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, size_t n) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (; i + 3 < n; i += stride) *reinterpret_cast<float4*>(p + i) = float4{0.f, 0.f, 0.f, 0.f};
  if (i < n) for (size_t j = i; j < n && j < i + 4; ++j) p[j] = 0.f;
}
static int launch_zero_f32(float* p, size_t n, hipStream_t st) {
  if (n == 0) return SGPO_OK;
  SGPO_REQUIRE(((uintptr_t)p & 15) == 0, "zero_f32: buffer must be 16-byte aligned");
  size_t blocks = (n / 4 + 255) / 256 + 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, n);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

static int guidance_ratio(sgpo_mlp* m, const int64_t* x, float t, int B, int L, int S, int mask_index, float* log_ratio,
                          void* workspace, size_t workspace_bytes, void* stream) {
  SGPO_REQUIRE(m && x && log_ratio && workspace, "guidance_ratio_mlp: null argument");
  SGPO_REQUIRE(B > 0 && L > 0 && S > 0 && S <= 32 && S <= m->alphabet, "guidance_ratio_mlp: B=%d L=%d S=%d (S <= min(32, alphabet))", B, L, S);
  SGPO_REQUIRE(mask_index < S, "guidance_ratio_mlp: mask_index %d outside the %d target states", mask_index, S);
  SGPO_REQUIRE(m->n_read < 65536, "guidance_ratio_mlp: n_read %d", m->n_read);
  if (int rc = check_read_pos(m, L, "guidance_ratio_mlp")) return rc;
  if (workspace_bytes < sgpo_mlp_workspace_bytes(m, B)) {
    sgpo_set_error("guidance_ratio_mlp: workspace %zu < required %zu bytes", workspace_bytes, sgpo_mlp_workspace_bytes(m, B));
    return SGPO_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* w = (char*)workspace;
  float* hpre = (float*)w;
  int* cnt = (int*)(w + al256((size_t)B * MH * sizeof(float)));
  unsigned short* lists = (unsigned short*)((char*)cnt + al256((size_t)B * 2 * sizeof(int)));
  const bool compact = mask_index >= 0;
  if (compact)
    if (int rc = launch_zero_f32(log_ratio, (size_t)B * m->n_read * S, st)) return rc;
  {
    ProfScope prof_scope(PK_MLP_BASE, st, (double)B * m->n_read * MH * 4.0);
    hipLaunchKernelGGL(mlp_base_kernel, dim3(B), dim3(64), 0, st, mlp_dev(m), x, (const float*)nullptr, t, L, hpre,
                       (float*)nullptr, mask_index, compact ? cnt : (int*)nullptr, compact ? lists : (unsigned short*)nullptr);
  }
  const int spread = (compact && B < 1024) ? 1 : 0;   // few sequences: one tile slot per wave instead of one sequence
  const int n_units = (compact && !spread) ? B : B * m->n_read;
  int blocks = (n_units + 3) / 4;
  if (blocks > 256 * 3) blocks = 256 * 3;  // 3 resident blocks per CU (LDS 49 KiB each), grid-stride the rest
  // b5 is added to both f and fbase and cancels in the ratio.
  const float b5 = 0.f;
  // factored jump cost: three 64x64 layers + first-layer update + head per (b, i, s).  In compact mode the
  // work depends on the data; the figure reported is its lower bound (every position unmasked: one jump each)
  const double per_jump = 3.0 * 2 * MH * MH + 5.0 * MH;
  ProfScope prof_scope(PK_MLP_JUMP, st, compact ? (double)B * m->n_read * per_jump : (double)B * m->n_read * S * per_jump);
  hipLaunchKernelGGL(mlp_jump_kernel, dim3(blocks), dim3(256), 0, st, m->w_perm, m->bias_perm, m->T1, m->read_pos, hpre, x,
                     b5, B, m->n_read, m->alphabet, S, L, mask_index, compact ? cnt : (const int*)nullptr,
                     compact ? lists : (const unsigned short*)nullptr, spread, log_ratio);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

extern "C" int sgpo_guidance_ratio_mlp(sgpo_mlp* m, const int64_t* x, float t, int B, int L, int S, float* log_ratio,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  return guidance_ratio(m, x, t, B, L, S, -1, log_ratio, workspace, workspace_bytes, stream);
}

extern "C" int sgpo_guidance_ratio_mlp_absorbing(sgpo_mlp* m, const int64_t* x, float t, int B, int L, int S, int mask_index,
                                                 float* log_ratio, void* workspace, size_t workspace_bytes, void* stream) {
  SGPO_REQUIRE(mask_index >= 0, "guidance_ratio_mlp_absorbing: mask_index %d", mask_index);
  return guidance_ratio(m, x, t, B, L, S, mask_index, log_ratio, workspace, workspace_bytes, stream);
}

// ---- one guided step for several stacked sampler calls -------------------------------------------------------------
// Rows [row_lo[g], row_lo[g + 1]) of the batch belong to draw g, which has its own value function and its own Philox
// key.  The kernels of sgpo_guidance_ratio_mlp_absorbing and sgpo_reverse_step_mdlm run ONCE for up to SGPO_MAX_GROUPS
// draws (blockIdx.y picks the draw; inside a draw every index is that of its own launch, so the tokens are the same bit
// for bit) instead of four small launches per draw: ten draws of ten sequences were forty launches a step, a quarter of
// the step's GPU time and most of its launch gaps.
static int guided_step_groups_impl(int n_groups, sgpo_mlp* const* members, const int* row_lo, const float* logits,
                                   const int64_t* x, int L, int K, int S, float t, int mask_index,
                                   float move_chance_t, float move_chance_s, float guide_temp,
                                   const uint64_t* seeds, const uint64_t* offsets, const uint64_t* row_offsets,
                                   const int* slot_of_pos, float* log_ratio, size_t log_ratio_bytes,
                                   void* workspace, size_t workspace_bytes, int64_t* x_out, int* nan_flag,
                                   const uint64_t* keys_dev, void* stream) {
  static const uint64_t zeros[1024] = {0};
  if (keys_dev) {   // the keys come from device memory: the host arrays are not read (seed, first row), offsets[] may be null
    SGPO_REQUIRE(n_groups <= 1024, "guided_step_groups_devkey: %d draws (at most 1024)", n_groups);
    seeds = zeros; row_offsets = zeros;
    if (!offsets) offsets = zeros;
  }
  SGPO_REQUIRE(n_groups > 0 && members && row_lo && logits && x && seeds && offsets && row_offsets && x_out,
               "guided_step_groups: null argument");
  SGPO_REQUIRE(L > 0 && K > 0 && row_lo[0] >= 0, "guided_step_groups: L=%d K=%d first row %d", L, K, row_lo[0]);
  hipStream_t st = (hipStream_t)stream;
  const int B = row_lo[n_groups];
  // the value functions of one call share the read positions (one slot map, one workspace layout over all B rows)
  const sgpo_mlp* first = nullptr;
  for (int g = 0; g < n_groups; ++g) {
    SGPO_REQUIRE(row_lo[g + 1] > row_lo[g], "guided_step_groups: group %d has %d rows", g, row_lo[g + 1] - row_lo[g]);
    const sgpo_mlp* m = (guide_temp != 0.f) ? members[g] : nullptr;
    if (!m) continue;
    if (!first) first = m;
    SGPO_REQUIRE(m->n_read == first->n_read && m->alphabet == first->alphabet, "guided_step_groups: group %d reads %d positions over %d letters, group 0 %d over %d",
                 g, m->n_read, m->alphabet, first->n_read, first->alphabet);
    for (int i = 0; i < m->n_read; ++i)
      SGPO_REQUIRE(m->read_pos_host[i] == first->read_pos_host[i], "guided_step_groups: group %d reads other positions than the first", g);
  }
  float* hpre = nullptr; int* cnt = nullptr; unsigned short* lists = nullptr;
  if (first) {
    SGPO_REQUIRE(log_ratio && workspace && mask_index >= 0 && mask_index < S && S > 0 && S <= 32 && S <= first->alphabet,
                 "guided_step_groups: log_ratio/workspace missing or S=%d mask %d", S, mask_index);
    SGPO_REQUIRE(first->n_read < 65536, "guided_step_groups: n_read %d", first->n_read);
    if (int rc = check_read_pos(first, L, "guided_step_groups")) return rc;
    const size_t need_lr = (size_t)B * first->n_read * S * sizeof(float), need_ws = sgpo_mlp_workspace_bytes(first, B);
    if (log_ratio_bytes < need_lr || workspace_bytes < need_ws) {
      sgpo_set_error("guided_step_groups: log_ratio %zu / workspace %zu bytes, need %zu / %zu for %d rows", log_ratio_bytes,
                     workspace_bytes, need_lr, need_ws, B);
      return SGPO_ERR_WORKSPACE;
    }
    char* w = (char*)workspace;
    hpre = (float*)w;
    cnt = (int*)(w + al256((size_t)B * MH * sizeof(float)));
    lists = (unsigned short*)((char*)cnt + al256((size_t)B * 2 * sizeof(int)));
    if (int rc = launch_zero_f32(log_ratio, need_lr / sizeof(float), st)) return rc;
  }
  for (int g0 = 0; g0 < n_groups; g0 += SGPO_MAX_GROUPS) {
    const int ng = n_groups - g0 < SGPO_MAX_GROUPS ? n_groups - g0 : SGPO_MAX_GROUPS;
    RsGroups rs{};
    MlpGroups mg{};
    JumpGroups jg{};
    rs.n = ng;
    int max_seq = 0, max_guided = 0;
    double base_work = 0, jump_work = 0;
    for (int i = 0; i < ng; ++i) {
      const int g = g0 + i, lo = row_lo[g], n = row_lo[g + 1] - lo;
      const sgpo_mlp* m = (guide_temp != 0.f) ? members[g] : nullptr;
      rs.g[i] = RsGroup{lo, n, m ? 1 : 0, 0, seeds[g], offsets[g], row_offsets[g]};
      mg.seq_lo[i] = jg.seq_lo[i] = lo;
      mg.n_seq[i] = jg.n_seq[i] = m ? n : 0;
      if (m) {
        mg.d[i] = mlp_dev(m);
        jg.w_perm[i] = m->w_perm; jg.bias_perm[i] = m->bias_perm; jg.T1[i] = m->T1;
        if (n > max_guided) max_guided = n;
        base_work += (double)n * m->n_read * MH * 4.0;
        jump_work += (double)n * m->n_read * (3.0 * 2 * MH * MH + 5.0 * MH);
      }
      if (n > max_seq) max_seq = n;
    }
    if (max_guided > 0) {
      {
        ProfScope prof_scope(PK_MLP_BASE, st, base_work);
        hipLaunchKernelGGL(mlp_base_groups_kernel, dim3(max_guided, ng), dim3(64), 0, st, mg, x, t, L, hpre, mask_index, cnt, lists);
      }
      const int spread = max_guided < 1024 ? 1 : 0;
      const int n_units = spread ? max_guided * first->n_read : max_guided;
      int blocks = (n_units + 3) / 4;
      const int cap = (256 * 3 + ng - 1) / ng;   // about three resident blocks per CU over all draws, grid-stride the rest
      if (blocks > cap) blocks = cap;
      ProfScope prof_scope(PK_MLP_JUMP, st, jump_work);
      hipLaunchKernelGGL(mlp_jump_groups_kernel, dim3(blocks, ng), dim3(256), 0, st, jg, first->read_pos, hpre, x,
                         first->n_read, first->alphabet, S, L, mask_index, cnt, lists, log_ratio);
    }
    SGPO_CHECK_HIP(hipGetLastError());
    if (int rc = launch_reverse_step_mdlm_groups(rs, max_seq, logits, x, first ? log_ratio : nullptr, first ? slot_of_pos : nullptr,
                                                 first ? first->n_read : L, L, K, mask_index, move_chance_t, move_chance_s,
                                                 guide_temp, x_out, nan_flag, st, keys_dev, g0))
      return rc;
  }
  return SGPO_OK;
}

extern "C" int sgpo_guided_step_groups_mdlm(int n_groups, sgpo_mlp* const* members, const int* row_lo, const float* logits,
                                            const int64_t* x, int L, int K, int S, float t, int mask_index,
                                            float move_chance_t, float move_chance_s, float guide_temp,
                                            const uint64_t* seeds, const uint64_t* offsets, const uint64_t* row_offsets,
                                            const int* slot_of_pos, float* log_ratio, size_t log_ratio_bytes,
                                            void* workspace, size_t workspace_bytes, int64_t* x_out, int* nan_flag,
                                            void* stream) {
  return guided_step_groups_impl(n_groups, members, row_lo, logits, x, L, K, S, t, mask_index, move_chance_t, move_chance_s,
                                 guide_temp, seeds, offsets, row_offsets, slot_of_pos, log_ratio, log_ratio_bytes, workspace,
                                 workspace_bytes, x_out, nan_flag, nullptr, stream);
}

extern "C" int sgpo_guided_step_groups_mdlm_devkey(int n_groups, sgpo_mlp* const* members, const int* row_lo,
                                                   const float* logits, const int64_t* x, int L, int K, int S, float t,
                                                   int mask_index, float move_chance_t, float move_chance_s, float guide_temp,
                                                   uint64_t offset, const uint64_t* keys_dev, const int* slot_of_pos,
                                                   float* log_ratio, size_t log_ratio_bytes, void* workspace,
                                                   size_t workspace_bytes, int64_t* x_out, int* nan_flag, void* stream) {
  SGPO_REQUIRE(keys_dev != nullptr && n_groups > 0 && n_groups <= 1024, "guided_step_groups_devkey: null keys or %d draws", n_groups);
  uint64_t offs[1024];
  for (int g = 0; g < n_groups; ++g) offs[g] = offset;
  return guided_step_groups_impl(n_groups, members, row_lo, logits, x, L, K, S, t, mask_index, move_chance_t, move_chance_s,
                                 guide_temp, nullptr, offs, nullptr, slot_of_pos, log_ratio, log_ratio_bytes, workspace,
                                 workspace_bytes, x_out, nan_flag, keys_dev, stream);
}

// ---- the same call for callers that cannot allocate -----------------------------------------
// The workspace of guidance_ratio (hpre | cnt | lists) lives in a device array that is part of the code
// object: every GPU the library runs on holds its own copy, at its own address, until the process ends.
// `scratch` is only the host-side name of that array; the device address is asked for once per device
// (sgpo_guidance_init_device, called by sgpo_init, or the first call on a device) and host code uses that
// pointer from then on.
static __device__ __align__(256) unsigned char scratch[SGPO_GUIDANCE_NOALLOC_MAX_BYTES];
#define SGPO_MAX_DEVICES 16
static void* g_scratch_ptr[SGPO_MAX_DEVICES] = {nullptr};
static std::mutex g_scratch_mutex;

static int scratch_of_current_device(void** out) {
  int dev = -1;
  SGPO_CHECK_HIP(hipGetDevice(&dev));
  SGPO_REQUIRE(dev >= 0 && dev < SGPO_MAX_DEVICES, "guidance scratch: device index %d outside [0,%d)", dev, SGPO_MAX_DEVICES);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    SGPO_CHECK_HIP(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
    SGPO_REQUIRE(p != nullptr, "guidance scratch: hipGetSymbolAddress returned a null address on device %d", dev);
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return SGPO_OK;
}

extern "C" int sgpo_guidance_init_device() {  // called from sgpo_init with the rank's device current
  void* p = nullptr;
  return scratch_of_current_device(&p);
}

extern "C" size_t sgpo_guidance_noalloc_max_bytes(void) { return SGPO_GUIDANCE_NOALLOC_MAX_BYTES; }

extern "C" int sgpo_guidance_ratio_mlp_absorbing_noalloc(sgpo_mlp* m, const int64_t* x, float t, int B, int L, int S,
                                                         int mask_index, float* log_ratio, void* stream) {
  SGPO_REQUIRE(m && x && log_ratio, "guidance_ratio_mlp_absorbing_noalloc: null argument");
  SGPO_REQUIRE(mask_index >= 0, "guidance_ratio_mlp_absorbing_noalloc: mask_index %d", mask_index);
  SGPO_REQUIRE(B > 0, "guidance_ratio_mlp_absorbing_noalloc: B=%d", B);
  const size_t need = sgpo_mlp_workspace_bytes(m, B);
  if (need > SGPO_GUIDANCE_NOALLOC_MAX_BYTES) {
    sgpo_set_error("guidance_ratio_mlp_absorbing_noalloc: B=%d with %d read positions needs %zu bytes of scratch, the built-in "
                   "buffer holds %zu; use sgpo_guidance_ratio_mlp_absorbing with a caller workspace", B, m->n_read, need,
                   (size_t)SGPO_GUIDANCE_NOALLOC_MAX_BYTES);
    return SGPO_ERR_WORKSPACE;
  }
  void* ws = nullptr;
  if (int rc = scratch_of_current_device(&ws)) return rc;
  // nothing in the buffer has to start at zero: mlp_base_kernel writes hpre, cnt and lists of every row b < B before
  // mlp_jump_kernel (same stream) reads them, and no row >= B is read
  return guidance_ratio(m, x, t, B, L, S, mask_index, log_ratio, ws, need, stream);
}
