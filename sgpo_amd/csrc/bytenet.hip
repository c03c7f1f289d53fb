// ByteNetLMTime denoiser of the D3PM path (third party `evodiff.model.ByteNetLMTime`, restated
// from the published architecture -- SURVEY.md Appendix A; call site models/d3pm.py:34-37;
// hyper-parameters configs/pretrain_model/d3pm.yaml:12-27).  PARITY UNPINNED: checked against the
// oracle's restatement (oracle/bytenet.py) only.
//
// Per layer:  x += PFF2( GELU(LN( DilatedConv5( GELU(LN( PFF1( GELU(LN(x)) ) )) ) )) )
// The two position-wise feed-forwards and the dilated convolution are bf16 MFMA GEMMs; the
// convolution is ONE implicit GEMM with K = kernel_size * d_h whose K-segments read the operand
// rows shifted by (tap - k/2) * dilation inside each sequence (zero rows outside), so the
// [M][5*512] im2col matrix is never materialised.  LayerNorm (affine) + exact GELU are fused
// in one HBM-bound pass that also rounds to bf16 for the next GEMM.
#include <math.h>
#include <stdlib.h>

#include <vector>

#include "../../include/sgpo_hip.h"
#include "dit.h"
#include "gemm.h"
#include "prof.h"

int launch_gemm_bf16_seg(const GemmParams& p, int epi, int n_seg, int seg_k, const int* shifts, hipStream_t stream);

static int launch_ln_gelu(const float* x, const float* w, const float* b, bf16_t* out, int M, int H, hipStream_t st);

// GEMM -> LayerNorm -> GELU -> bf16.  Big batches take the full-row persistent kernel that normalises in its epilogue
// (gemm_p.hip, EPI_LN_GELU: the fp32 pre-LayerNorm activations never reach HBM); small ones, other widths, or
// SGPO_BYTENET_FUSED_LN=0 (A/B measurements, tests) take the fp32-output GEMM and the separate pass.
// The fused kernel writes `out_fused` (it must not alias the operand: other tiles still read it); the two-pass form
// writes fp32 to `c32` and then bf16 to `out_pass` (which may be the operand's buffer).  *used = where the result is.
static int gemm_ln_gelu(GemmParams g, const float* ln_w, const float* ln_b, float* c32, bf16_t* out_fused, bf16_t* out_pass,
                        const bf16_t** used, int n_seg, int seg_k, const int* shifts, hipStream_t st) {
  const char* e = getenv("SGPO_BYTENET_FUSED_LN");
  if (!(e && e[0] == '0')) {
    GemmParams f = g;
    f.C = out_fused; f.ln_w = ln_w; f.ln_b = ln_b; f.ln_eps = 1e-5f;
    const int rc = launch_gemm_ln_gelu(f, n_seg, seg_k, shifts, st);
    if (rc != SGPO_ERR_UNSUPPORTED) { *used = out_fused; return rc; }
  }
  g.C = c32;
  int rc = launch_gemm_bf16_seg(g, EPI_BIAS_F32, n_seg, seg_k, shifts, st);
  if (rc) return rc;
  *used = out_pass;
  return launch_ln_gelu(c32, ln_w, ln_b, out_pass, g.M, g.N, st);
}

struct BnLayer {
  const float *ln1_w, *ln1_b, *b1, *ln2_w, *ln2_b, *bc, *ln3_w, *ln3_b, *b2;
  const bf16_t *w1, *wc, *w2;
  int dilation;
};

struct sgpo_bytenet {
  sgpo_bytenet_config cfg;
  float* f32 = nullptr;
  bf16_t* b16 = nullptr;
  const float *embed, *up_w, *up_b, *dec_b;
  const bf16_t* dec_w;
  std::vector<BnLayer> layers;
};

static size_t bn_layer_params(const sgpo_bytenet_config& c) {
  const size_t D = c.d_model, H = c.d_h, K = c.kernel_size;
  return 2 * D + H * D + H + 2 * H + H * H * K + H + 2 * H + D * H + D;
}
extern "C" size_t sgpo_bytenet_param_count(const sgpo_bytenet_config* c) {
  if (!c) return 0;
  const size_t D = c->d_model, E = c->d_embedding, V = c->n_tokens;
  return V * E + D * E + D + (size_t)c->n_layers * bn_layer_params(*c) + V * D + V;
}

// conv weight [out][in][tap] (torch Conv1d) -> bf16 [out][tap][in]
__global__ void repack_conv_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int H, int K) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)H * H * K) return;
  const int in = idx % H, tap = (idx / H) % K, out = idx / ((size_t)H * K);
  dst[idx] = f2bf(src[((size_t)out * H + in) * K + tap]);
}

// x0[m][:] = W_up (embed[token] + pe[t_b]) + b_up    (K = d_embedding = 8: VALU)
__global__ __launch_bounds__(256) void bn_embed_kernel(const int64_t* __restrict__ tok, const int64_t* __restrict__ t,
                                                       int n_t, const float* __restrict__ emb, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ x, int M, int L,
                                                       int D, int E, int V) {
  const int row = blockIdx.x;
  if (row >= M) return;
  __shared__ float e[64];
  if (threadIdx.x < E) {
    int64_t tk = tok[row];
    tk = tk < 0 ? 0 : (tk >= V ? V - 1 : tk);
    const float tv = (float)t[n_t == 1 ? 0 : row / L];
    const int i = threadIdx.x;
    const float div = expf((float)(i & ~1) * (-logf(10000.0f) / (float)E));
    const float pe = (i & 1) ? cosf(tv * div) : sinf(tv * div);
    e[i] = emb[tk * E + i] + pe;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < D; j += 256) {
    float a = b[j];
    for (int k = 0; k < E; ++k) a = fmaf(w[j * E + k], e[k], a);
    x[(size_t)row * D + j] = a;
  }
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }

// out = bf16( GELU( LN(x) * w + b ) ), one wave per row, H in {512, 1024} register-resident, else strided
template <int H>
__global__ __launch_bounds__(256) void ln_gelu_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ b, bf16_t* __restrict__ out, int M) {
  constexpr int PER = H / 64;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  float v[PER];
  const float* xr = x + (size_t)row * H;
#pragma unroll
  for (int i = 0; i < PER / 4; ++i) {
    float4 t = reinterpret_cast<const float4*>(xr)[i * 64 + lane];
    v[i * 4 + 0] = t.x; v[i * 4 + 1] = t.y; v[i * 4 + 2] = t.z; v[i * 4 + 3] = t.w;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) s += v[i];
  const float mean = wave_sum(s) * (1.0f / H);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) { float d = v[i] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / H) + 1e-5f);
#pragma unroll
  for (int i = 0; i < PER / 4; ++i) {
    const int col = (i * 64 + lane) * 4;
    float4 ww = *reinterpret_cast<const float4*>(w + col);
    float4 bb = *reinterpret_cast<const float4*>(b + col);
    float o0 = gelu_erf((v[i * 4 + 0] - mean) * rstd * ww.x + bb.x);
    float o1 = gelu_erf((v[i * 4 + 1] - mean) * rstd * ww.y + bb.y);
    float o2 = gelu_erf((v[i * 4 + 2] - mean) * rstd * ww.z + bb.z);
    float o3 = gelu_erf((v[i * 4 + 3] - mean) * rstd * ww.w + bb.w);
    uint2 pk;
    pk.x = pack_bf16x2(o0, o1);
    pk.y = pack_bf16x2(o2, o3);
    *reinterpret_cast<uint2*>(out + (size_t)row * H + col) = pk;
  }
}
// H = 512 in the arithmetic of the fused GEMM epilogue (gemm_p.hip, EPI_LN_GELU), operation for operation, so that a
// batch below the fused kernel's row threshold (GEMM + this pass) gives the bits of a batch above it: a rank's shard of
// 32 sequences must reproduce the rows of the 64-sequence call.  The epilogue's order: the row is cut
// into eight 64-column segments (its eight waves); in a segment, lane group e4 < 4 owns the quads at columns
// 32 nt + 16 nh + 4 e4 (nt < 2, nh < 2): quad sums accumulated over (nt, nh), the groups added as (0 + 2) + (1 + 3), / 64 ->
// mean_s; squared deviations from mean_s as one fma chain over (nt, nh, q), the groups added the same way -> M2_s; the
// eight (mean_s, M2_s) combined in segment order (Chan et al.); then fma(((v - mean) rstd), w, b), Abramowitz-Stegun
// GELU, bf16.  Here 16 lanes take a row (lane = 2 segment + h, holding groups e4 = h and h + 2), four rows per wave.
__global__ __launch_bounds__(256) void ln_gelu_row512_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, bf16_t* __restrict__ out, int M) {
  const int lane = threadIdx.x & 63;
  int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const bool live = row < M;
  if (!live) row = M - 1;                       // keep the shuffles whole; nothing is stored
  const int seg = (lane & 15) >> 1, h = lane & 1;
  const float* xr = x + (size_t)row * 512 + seg * 64 + 4 * h;
  float4 v[2][2][2];   // [e: group h / h + 2][nt][nh]
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) v[e][nt][nh] = *reinterpret_cast<const float4*>(xr + 8 * e + nt * 32 + 16 * nh);
  float pm2[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    float a = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
        a = rn_add(a, rn_add(rn_add(v[e][nt][nh].x, v[e][nt][nh].y), rn_add(v[e][nt][nh].z, v[e][nt][nh].w)));
    pm2[e] = a;
  }
  float t = rn_add(pm2[0], pm2[1]);                                         // group h + group h + 2
  float mean = rn_mul(rn_add(__shfl(t, lane & ~1), __shfl(t, lane | 1)), 1.0f / 64.0f);   // (0 + 2) + (1 + 3)
  float ps2[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    float sq = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        const float el[4] = {v[e][nt][nh].x, v[e][nt][nh].y, v[e][nt][nh].z, v[e][nt][nh].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float d = el[q] - mean; sq = fmaf(d, d, sq); }
      }
    ps2[e] = sq;
  }
  t = rn_add(ps2[0], ps2[1]);
  const float sq = rn_add(__shfl(t, lane & ~1), __shfl(t, lane | 1));
  float ms = 0.f, m2 = 0.f, pm[8];
#pragma unroll
  for (int s8 = 0; s8 < 8; ++s8) {
    pm[s8] = __shfl(mean, (lane & 48) + 2 * s8);
    ms = rn_add(ms, pm[s8]);
    m2 = rn_add(m2, __shfl(sq, (lane & 48) + 2 * s8));
  }
  const float mu = rn_mul(ms, 0.125f);
  float dv = 0.f;
#pragma unroll
  for (int s8 = 0; s8 < 8; ++s8) { const float d = pm[s8] - mu; dv = fmaf(d, d, dv); }
  m2 = fmaf(64.0f, dv, m2);
  const float rstd = rsqrtf(fmaf(m2, 1.0f / 512.0f, 1e-5f));
  if (!live) return;
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        const int col = seg * 64 + nt * 32 + 16 * nh + 8 * e + 4 * h;
        const float4 ww = *reinterpret_cast<const float4*>(w + col);
        const float4 bb = *reinterpret_cast<const float4*>(b + col);
        const float4 vv = v[e][nt][nh];
        const float o0 = epi_gelu_erf_fast(fmaf(rn_mul(vv.x - mu, rstd), ww.x, bb.x));
        const float o1 = epi_gelu_erf_fast(fmaf(rn_mul(vv.y - mu, rstd), ww.y, bb.y));
        const float o2 = epi_gelu_erf_fast(fmaf(rn_mul(vv.z - mu, rstd), ww.z, bb.z));
        const float o3 = epi_gelu_erf_fast(fmaf(rn_mul(vv.w - mu, rstd), ww.w, bb.w));
        uint2 pk;
        pk.x = pack_bf16x2(o0, o1);
        pk.y = pack_bf16x2(o2, o3);
        *reinterpret_cast<uint2*>(out + (size_t)row * 512 + col) = pk;
      }
}
__global__ __launch_bounds__(256) void ln_gelu_generic_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, bf16_t* __restrict__ out, int M,
                                                              int H) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + (size_t)row * H;
  float s = 0.f;
  for (int i = lane; i < H; i += 64) s += xr[i];
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
  for (int i = lane; i < H; i += 64) { float d = xr[i] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + 1e-5f);
  for (int i = lane; i < H; i += 64) out[(size_t)row * H + i] = f2bf(gelu_erf((xr[i] - mean) * rstd * w[i] + b[i]));
}
static int launch_ln_gelu(const float* x, const float* w, const float* b, bf16_t* out, int M, int H, hipStream_t st) {
  dim3 grid((M + 3) / 4), block(256);
  ProfScope prof_scope(PK_LN_MOD, st, (double)M * H * 6.0);
  if (H == 512) hipLaunchKernelGGL(ln_gelu_row512_kernel, dim3((M + 15) / 16), block, 0, st, x, w, b, out, M);
  else if (H == 1024) hipLaunchKernelGGL(ln_gelu_kernel<1024>, grid, block, 0, st, x, w, b, out, M);
  else hipLaunchKernelGGL(ln_gelu_generic_kernel, grid, block, 0, st, x, w, b, out, M, H);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

extern "C" void sgpo_bytenet_destroy(sgpo_bytenet* m) {
  if (!m) return;
  (void)hipFree(m->f32); (void)hipFree(m->b16);
  delete m;
}

extern "C" int sgpo_bytenet_create(const sgpo_bytenet_config* cfg, const float* blob, size_t n_floats, sgpo_bytenet** out) {
  SGPO_REQUIRE(cfg && blob && out, "bytenet_create: null argument");
  const sgpo_bytenet_config c = *cfg;
  SGPO_REQUIRE(c.n_tokens > 0 && c.n_tokens <= 128 && c.d_embedding > 0 && c.d_embedding <= 64 && c.d_embedding % 2 == 0,
               "bytenet_create: n_tokens=%d d_embedding=%d", c.n_tokens, c.d_embedding);
  SGPO_REQUIRE(c.d_model % 64 == 0 && c.d_h % 64 == 0 && c.n_layers > 0 && c.kernel_size > 0 && c.kernel_size <= 8 &&
               (c.kernel_size & 1) && c.r >= 1, "bytenet_create: d_model/d_h must be multiples of 64, odd kernel_size <= 7");
  SGPO_REQUIRE(n_floats == sgpo_bytenet_param_count(cfg), "bytenet_create: blob has %zu floats, expected %zu", n_floats,
               sgpo_bytenet_param_count(cfg));
  sgpo_bytenet* m = new sgpo_bytenet();
  m->cfg = c;
  const size_t D = c.d_model, H = c.d_h, K = c.kernel_size, E = c.d_embedding, V = c.n_tokens;
  hipStream_t st = 0;
#define HIPF(e) do { hipError_t _e = (e); if (_e != hipSuccess) { sgpo_set_error("%s: %s", #e, hipGetErrorString(_e)); sgpo_bytenet_destroy(m); return SGPO_ERR_HIP; } } while (0)
  HIPF(hipMalloc((void**)&m->f32, n_floats * 4));
  HIPF(hipMemcpyAsync(m->f32, blob, n_floats * 4, hipMemcpyDeviceToDevice, st));
  const size_t n16 = (size_t)c.n_layers * (H * D + H * H * K + D * H) + V * D;
  HIPF(hipMalloc((void**)&m->b16, n16 * sizeof(bf16_t)));
  const float* p = m->f32;
  bf16_t* q = m->b16;
  auto take = [&](size_t n) { const float* r = p; p += n; return r; };
  int rc = 0;
  auto take16 = [&](size_t n) {
    const float* src = take(n);
    bf16_t* dst = q; q += n;
    int r = launch_f32_to_bf16(src, dst, n, st);
    if (r) rc = r;
    return (const bf16_t*)dst;
  };
  m->embed = take(V * E); m->up_w = take(D * E); m->up_b = take(D);
  const int log2r = (int)floor(log2((double)c.r)) + 1;
  m->layers.resize(c.n_layers);
  for (int n = 0; n < c.n_layers; ++n) {
    BnLayer& l = m->layers[n];
    l.dilation = 1 << (n % log2r);
    l.ln1_w = take(D); l.ln1_b = take(D);
    l.w1 = take16(H * D); l.b1 = take(H);
    l.ln2_w = take(H); l.ln2_b = take(H);
    {
      const float* src = take(H * H * K);
      bf16_t* dst = q; q += H * H * K;
      const size_t nel = H * H * K;
      hipLaunchKernelGGL(repack_conv_kernel, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, src, dst, (int)H, (int)K);
      l.wc = dst;
    }
    l.bc = take(H);
    l.ln3_w = take(H); l.ln3_b = take(H);
    l.w2 = take16(D * H); l.b2 = take(D);
  }
  m->dec_w = take16(V * D); m->dec_b = take(V);
  if (rc) { sgpo_bytenet_destroy(m); return rc; }
  HIPF(hipGetLastError());
  HIPF(hipStreamSynchronize(st));
#undef HIPF
  *out = m;
  return SGPO_OK;
}

struct BnWs { float* x; bf16_t* a; float* c; bf16_t* d; size_t total; };
static BnWs bn_carve(const sgpo_bytenet* m, int B, int L, char* base) {
  const size_t M = (size_t)B * L, D = m->cfg.d_model, H = m->cfg.d_h;
  BnWs w; size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return base ? base + o : (char*)nullptr; };
  w.x = (float*)carve(M * D * 4);
  w.a = (bf16_t*)carve(M * D * 2);
  w.c = (float*)carve(M * H * 4);
  w.d = (bf16_t*)carve(M * H * 2);
  w.total = off;
  return w;
}
extern "C" size_t sgpo_bytenet_workspace_bytes(const sgpo_bytenet* m, int B, int L) {
  if (!m || B <= 0 || L <= 0) return 0;
  return bn_carve(m, B, L, nullptr).total;
}

extern "C" int sgpo_bytenet_forward(sgpo_bytenet* m, const int64_t* tokens, const int64_t* t, int n_t, int B, int L,
                                    float* logits, float* hidden_out, void* workspace, size_t workspace_bytes,
                                    void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  SGPO_REQUIRE(m && tokens && t && logits && workspace, "bytenet_forward: null argument");
  SGPO_REQUIRE(B > 0 && L > 0 && (n_t == 1 || n_t == B), "bytenet_forward: B=%d L=%d n_t=%d (1 or B)", B, L, n_t);
  SGPO_REQUIRE(((uintptr_t)workspace & 255) == 0, "bytenet_forward: workspace must be 256-B aligned");
  if (workspace_bytes < sgpo_bytenet_workspace_bytes(m, B, L)) {
    sgpo_set_error("bytenet_forward: workspace %zu < required %zu bytes", workspace_bytes, sgpo_bytenet_workspace_bytes(m, B, L));
    return SGPO_ERR_WORKSPACE;
  }
  const sgpo_bytenet_config& c = m->cfg;
  const int D = c.d_model, H = c.d_h, K = c.kernel_size, M = B * L;
  BnWs w = bn_carve(m, B, L, (char*)workspace);
  {
    ProfScope prof_scope(PK_EMBED, st, (double)M * D * 4.0);
    hipLaunchKernelGGL(bn_embed_kernel, dim3(M), dim3(256), 0, st, tokens, t, n_t, m->embed, m->up_w, m->up_b, w.x, M, L, D,
                       c.d_embedding, c.n_tokens);
  }
  int rc;
  for (const BnLayer& l : m->layers) {
    rc = launch_ln_gelu(w.x, l.ln1_w, l.ln1_b, w.a, M, D, st);
    if (rc) return rc;
    GemmParams g = {};
    g.A = w.a; g.W = l.w1; g.C = w.c; g.bias = l.b1; g.M = M; g.N = H; g.K = D; g.lda = D; g.ldw = D; g.ldc = H; g.L = L; g.B = B;
    const bf16_t *h1 = nullptr, *h2 = nullptr;
    rc = gemm_ln_gelu(g, l.ln2_w, l.ln2_b, w.c, w.d, w.d, &h1, 1, D, nullptr, st);
    if (rc) return rc;
    int shifts[8];
    for (int tap = 0; tap < K; ++tap) shifts[tap] = (tap - K / 2) * l.dilation;
    g = GemmParams{};
    g.A = h1; g.W = l.wc; g.C = w.c; g.bias = l.bc; g.M = M; g.N = H; g.K = K * H; g.lda = H; g.ldw = K * H; g.ldc = H;
    g.L = L; g.B = B;
    // fused: the bf16 result goes to the (then unused) fp32 scratch; two-pass: back into the operand's buffer
    rc = gemm_ln_gelu(g, l.ln3_w, l.ln3_b, w.c, reinterpret_cast<bf16_t*>(w.c), w.d, &h2, K, H, shifts, st);
    if (rc) return rc;
    g = GemmParams{};
    g.A = h2; g.W = l.w2; g.C = w.x; g.xin = w.x; g.bias = l.b2; g.M = M; g.N = D; g.K = H; g.lda = H; g.ldw = H; g.ldc = D;
    g.L = L; g.B = B;
    rc = launch_gemm_bf16(g, EPI_RES, st);
    if (rc) return rc;
  }
  if (hidden_out) SGPO_CHECK_HIP(hipMemcpyAsync(hidden_out, w.x, (size_t)M * D * 4, hipMemcpyDeviceToDevice, st));
  rc = launch_f32_to_bf16(w.x, w.a, (size_t)M * D, st);
  if (rc) return rc;
  GemmParams g = {};
  g.A = w.a; g.W = m->dec_w; g.C = logits; g.bias = m->dec_b; g.M = M; g.N = c.n_tokens; g.K = D; g.lda = D; g.ldw = D;
  g.ldc = c.n_tokens; g.L = L; g.B = B;
  rc = launch_gemm_bf16(g, EPI_BIAS_F32, st);
  if (rc) return rc;
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}
