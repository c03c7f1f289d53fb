// Host side of the MDLM denoiser: weight handle + the per-forward launch sequence.
// Reference: DIT.forward (dit.py:375-418), DDiTBlock.forward (dit.py:244-288),
// DDitFinalLayer.forward (dit.py:317-321).
#include <stdlib.h>
#include <stdarg.h>
#include <stdio.h>
#include <vector>

#include "../../include/sgpo_hip.h"
#include "dit.h"
#include "gemm.h"

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
extern "C" void sgpo_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* sgpo_last_error(void) { return g_err; }
extern "C" int sgpo_version(void) { return 1; }
extern "C" int sgpo_gemm_init_device();
extern "C" int sgpo_guidance_init_device();

// One device per process (one rank per GPU): the per-process caches of the launchers -- CU count and tile rule
// (gemm_p.hip), dynamic-LDS attributes (dit_kernels.hip, nos.hip) -- are taken on the device bound here, so a second
// device in the same process is refused instead of silently running with the first one's settings.
static int g_bound_device = -1;
extern "C" int sgpo_init(int device) {
  int n = 0;
  SGPO_CHECK_HIP(hipGetDeviceCount(&n));
  SGPO_REQUIRE(device >= 0 && device < n, "sgpo_init: device %d not in [0,%d)", device, n);
  SGPO_REQUIRE(g_bound_device < 0 || g_bound_device == device,
               "sgpo_init: this process is bound to device %d; the library supports one device per process (start one "
               "process per GPU)", g_bound_device);
  g_bound_device = device;
  SGPO_CHECK_HIP(hipSetDevice(device));
  if (int rc = sgpo_gemm_init_device()) return rc;
  return sgpo_guidance_init_device();   // device address of the allocation-free entry point's scratch array
}

// ------------------------------------------------------------------ small kernels
__global__ void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = f2bf(src[i]);
}
int launch_f32_to_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t stream) {
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) return SGPO_OK;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, dst, n);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}
extern "C" int sgpo_f32_to_bf16(const float* src, void* dst, size_t n, void* stream) {
  return launch_f32_to_bf16(src, (bf16_t*)dst, n, (hipStream_t)stream);
}

__global__ void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}

int launch_gemm_q(const GemmParams& p, int epi, hipStream_t stream);
// test / bench access to the two-workgroups-per-CU GEMM (gemm_q.hip): C bf16 [M][N] = epi(A W^T + bias), epi 0 = bias, 3 = gelu(tanh)
extern "C" int sgpo_gemm_q_bf16(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int epi, void* stream) {
  SGPO_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0, "gemm_q_bf16: bad arguments");
  GemmParams g = {};
  g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K;
  g.lda = K; g.ldw = K; g.ldc = N; g.L = M; g.B = 1;
  const int rc = launch_gemm_q(g, epi, (hipStream_t)stream);
  if (rc == SGPO_ERR_UNSUPPORTED) sgpo_set_error("gemm_q_bf16: shape M=%d N=%d K=%d epi=%d not supported", M, N, K, epi);
  return rc;
}

extern "C" int sgpo_gemm_bf16(const void* A, const void* W, float* C, int M, int N, int K, void* stream) {
  GemmParams p = {};
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.C = C;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.ldw = K; p.ldc = N; p.L = M > 0 ? M : 1; p.B = 1;
  return launch_gemm_bf16(p, EPI_BIAS_F32, (hipStream_t)stream);
}

// ------------------------------------------------------------------ model handle
#include "dit_model.h"

static size_t dit_block_params(const sgpo_dit_config& c) {
  const size_t H = c.hidden, F = c.mlp_hidden, C = c.cond_dim;
  return H + 3 * H * H + H * H + H + F * H + F + H * F + H + 6 * H * C + 6 * H;
}

extern "C" size_t sgpo_dit_param_count(const sgpo_dit_config* c) {
  if (!c) return 0;
  const size_t H = c->hidden, C = c->cond_dim, V = c->vocab, Fq = c->freq_dim;
  size_t n = V * H + C * Fq + C + C * C + C;
  n += (size_t)c->n_blocks * dit_block_params(*c);
  n += H + V * H + V + 2 * H * C + 2 * H;
  return n;
}

extern "C" void sgpo_dit_destroy(sgpo_dit* m) {
  if (!m) return;
  (void)hipFree(m->f32); (void)hipFree(m->b16); (void)hipFree(m->ada_w); (void)hipFree(m->ada_b);
  (void)hipFree(m->rope_cos); (void)hipFree(m->rope_sin);
  delete m;
}

extern "C" int sgpo_dit_create(const sgpo_dit_config* cfg, const float* blob, size_t n_floats, sgpo_dit** out) {
  SGPO_REQUIRE(cfg && blob && out, "dit_create: null argument");
  const sgpo_dit_config c = *cfg;
  SGPO_REQUIRE(c.hidden > 0 && c.n_heads > 0 && c.hidden == c.n_heads * 64,
               "dit_create: hidden %d / heads %d -> head_dim must be 64", c.hidden, c.n_heads);
  SGPO_REQUIRE(c.hidden % 128 == 0 && c.mlp_hidden % 128 == 0, "dit_create: hidden/mlp_hidden must be multiples of 128");
  SGPO_REQUIRE(c.vocab > 0 && c.vocab <= 128 && c.cond_dim > 0 && c.freq_dim > 0 && c.freq_dim % 2 == 0 && c.n_blocks > 0 &&
               c.max_len > 0, "dit_create: bad config");
  SGPO_REQUIRE(n_floats == sgpo_dit_param_count(cfg), "dit_create: blob has %zu floats, expected %zu", n_floats,
               sgpo_dit_param_count(cfg));
  sgpo_dit* m = new sgpo_dit();
  m->cfg = c;
  const size_t H = c.hidden, F = c.mlp_hidden, C = c.cond_dim, V = c.vocab, Fq = c.freq_dim;
  hipStream_t st = 0;
#define FAIL_IF(rc) do { int _r = (rc); if (_r) { sgpo_dit_destroy(m); return _r; } } while (0)
#define HIPF(e) do { hipError_t _e = (e); if (_e != hipSuccess) { sgpo_set_error("%s: %s", #e, hipGetErrorString(_e)); sgpo_dit_destroy(m); return SGPO_ERR_HIP; } } while (0)
  HIPF(hipMalloc((void**)&m->f32, n_floats * sizeof(float)));
  HIPF(hipMemcpyAsync(m->f32, blob, n_floats * sizeof(float), hipMemcpyDeviceToDevice, st));
  const size_t n_b16 = (size_t)c.n_blocks * (3 * H * H + H * H + 2 * F * H) + V * H;
  HIPF(hipMalloc((void**)&m->b16, n_b16 * sizeof(bf16_t)));
  m->n_mod = c.n_blocks * 6 * (int)H + 2 * (int)H;
  HIPF(hipMalloc((void**)&m->ada_w, (size_t)m->n_mod * C * sizeof(float)));
  HIPF(hipMalloc((void**)&m->ada_b, (size_t)m->n_mod * sizeof(float)));
  HIPF(hipMalloc((void**)&m->rope_cos, (size_t)c.max_len * 32 * sizeof(float)));
  HIPF(hipMalloc((void**)&m->rope_sin, (size_t)c.max_len * 32 * sizeof(float)));
  FAIL_IF(launch_rope_table(m->rope_cos, m->rope_sin, c.max_len, 32, st));

  const float* p = m->f32;
  bf16_t* q = m->b16;
  auto take = [&](size_t n) { const float* r = p; p += n; return r; };
  auto take16 = [&](size_t n, int* rc) {
    const float* src = take(n);
    bf16_t* dst = q; q += n;
    int r = launch_f32_to_bf16(src, dst, n, st);
    if (r) *rc = r;
    return (const bf16_t*)dst;
  };
  int rc = 0;
  m->embed = take(V * H);
  m->s_w1 = take(C * Fq); m->s_b1 = take(C); m->s_w2 = take(C * C); m->s_b2 = take(C);
  m->blocks.resize(c.n_blocks);
  for (int i = 0; i < c.n_blocks; ++i) {
    DitBlockW& b = m->blocks[i];
    b.norm1 = take(H);
    b.qkv = take16(3 * H * H, &rc);
    b.attn_out = take16(H * H, &rc);
    b.norm2 = take(H);
    b.fc1 = take16(F * H, &rc);
    b.fc1_b = take(F);
    b.fc2 = take16(H * F, &rc);
    b.fc2_b = take(H);
    const float* aw = take(6 * H * C);
    const float* ab = take(6 * H);
    HIPF(hipMemcpyAsync(m->ada_w + (size_t)i * 6 * H * C, aw, 6 * H * C * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPF(hipMemcpyAsync(m->ada_b + (size_t)i * 6 * H, ab, 6 * H * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  m->out_norm = take(H);
  m->out_lin_f32 = p;
  m->out_lin = take16(V * H, &rc);
  m->out_lin_b = take(V);
  {
    const float* aw = take(2 * H * C);
    const float* ab = take(2 * H);
    const size_t off = (size_t)c.n_blocks * 6 * H;
    HIPF(hipMemcpyAsync(m->ada_w + off * C, aw, 2 * H * C * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPF(hipMemcpyAsync(m->ada_b + off, ab, 2 * H * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  FAIL_IF(rc);
  HIPF(hipStreamSynchronize(st));
#undef FAIL_IF
#undef HIPF
  *out = m;
  return SGPO_OK;
}

// workspace carve-up (all offsets 256-B aligned)
struct DitWs {
  float* x; bf16_t* xn; bf16_t* qkv; bf16_t* att; bf16_t* mid; float* c; float* mod;
  size_t total;
};
static size_t al256(size_t n) { return (n + 255) & ~(size_t)255; }
static DitWs dit_carve(const sgpo_dit* m, int B, int L, char* base) {
  const size_t H = m->cfg.hidden, F = m->cfg.mlp_hidden;
  const size_t M = (size_t)B * L;
  DitWs w;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += al256(bytes); return base ? base + o : (char*)nullptr; };
  w.x = (float*)carve(M * H * 4);
  w.xn = (bf16_t*)carve(M * H * 2);
  w.qkv = (bf16_t*)carve(M * 3 * H * 2);
  w.att = (bf16_t*)carve(M * H * 2);
  w.mid = (bf16_t*)carve(M * F * 2);
  w.c = (float*)carve((size_t)B * m->cfg.cond_dim * 4);
  w.mod = (float*)carve((size_t)B * m->n_mod * 4);
  w.total = off;
  return w;
}

// (shift, scale) rows of the final layer's adaLN inside a forward workspace (valid after a forward)
const float* dit_ws_final_mod(const sgpo_dit* m, int B, int L, void* workspace) {
  DitWs w = dit_carve(m, B, L, (char*)workspace);
  return w.mod + (size_t)m->cfg.n_blocks * 6 * m->cfg.hidden;
}

// the fp32 residual stream after the last block (what hidden_out would receive), inside a forward workspace
const float* dit_ws_hidden(const sgpo_dit* m, int B, int L, void* workspace) {
  return dit_carve(m, B, L, (char*)workspace).x;
}

extern "C" size_t sgpo_dit_workspace_bytes(const sgpo_dit* m, int B, int L) {
  if (!m || B <= 0 || L <= 0) return 0;
  return dit_carve(m, B, L, nullptr).total;
}

extern "C" int sgpo_dit_forward(sgpo_dit* m, const int64_t* tokens, const float* sigma, int n_sigma, int B, int L,
                                const float* x_emb, float* logits, float* hidden_out, void* workspace,
                                size_t workspace_bytes, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  SGPO_REQUIRE(m && logits && workspace, "dit_forward: null argument");
  SGPO_REQUIRE(B > 0 && L > 0 && L <= m->cfg.max_len, "dit_forward: B=%d L=%d (max_len %d)", B, L, m->cfg.max_len);
  SGPO_REQUIRE(tokens || x_emb, "dit_forward: need tokens or x_emb");
  SGPO_REQUIRE(sigma == nullptr || n_sigma == 1 || n_sigma == B, "dit_forward: n_sigma must be 1 or B");
  SGPO_REQUIRE(((uintptr_t)workspace & 255) == 0, "dit_forward: workspace must be 256-B aligned");
  if (workspace_bytes < sgpo_dit_workspace_bytes(m, B, L)) {
    sgpo_set_error("dit_forward: workspace %zu < required %zu bytes", workspace_bytes, sgpo_dit_workspace_bytes(m, B, L));
    return SGPO_ERR_WORKSPACE;
  }
  const sgpo_dit_config& c = m->cfg;
  const int H = c.hidden, F = c.mlp_hidden, M = B * L;
  DitWs w = dit_carve(m, B, L, (char*)workspace);
  const int ns = sigma ? n_sigma : 1;
  const int mod_stride = (ns == 1) ? 0 : m->n_mod;
  int rc;
  // conditioning + every adaLN row in two launches
  rc = launch_conditioning(sigma, ns, m->s_w1, m->s_b1, m->s_w2, m->s_b2, c.freq_dim, c.cond_dim, w.c, m->ada_w,
                           m->ada_b, m->n_mod, w.mod, st);
  if (rc) return rc;

  const float* xcur;
  bool xn_ready = false;   // w.xn already holds the final layer's LayerNorm-modulate of xcur
  if (!x_emb) {
    // embedding gather and the first block's LayerNorm-modulate in one pass where the width allows
    rc = launch_embed_ln_mod(tokens, m->embed, c.vocab, w.x, m->blocks[0].norm1, w.mod, w.mod + H, mod_stride, w.xn, M, L, B, H, st);
    const bool first_ln_done = rc == SGPO_OK;
    if (rc == SGPO_ERR_UNSUPPORTED) rc = launch_embed(tokens, m->embed, w.x, M, H, c.vocab, st);
    if (rc) return rc;
    // out-projection / fc2 GEMM + gate + residual + the LayerNorm-modulate that follows in one kernel where the shape
    // allows (gemm_p.hip EPI_GATE_RES_LN: the separate HBM pass over the residual stream disappears); otherwise, or
    // with SGPO_DIT_FUSED_LN=0 (A/B measurements, tests), the GEMM and the pass.  Same bits either way (ln_stats.h).
    const char* fe = getenv("SGPO_DIT_FUSED_LN");
    const bool try_fused = !(fe && fe[0] == '0');
    auto gate_res_then_ln = [&](GemmParams g, const float* ln_w, const float* shift, const float* scale) -> int {
      if (try_fused) {
        GemmParams f = g;
        f.ln_w = ln_w; f.ln_shift = shift; f.ln_scale = scale; f.ln_out = w.xn; f.ln_eps = 1e-5f;
        const int r = launch_gemm_gate_res_ln(f, st);
        if (r != SGPO_ERR_UNSUPPORTED) return r;
      }
      const int r = launch_gemm_bf16(g, EPI_GATE_RES, st);
      if (r) return r;
      return launch_ln_mod(w.x, ln_w, shift, scale, mod_stride, w.xn, M, L, B, H, st);
    };
    for (int i = 0; i < c.n_blocks; ++i) {
      const DitBlockW& b = m->blocks[i];
      const float* mod = w.mod + (size_t)i * 6 * H;  // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
      if (i == 0 && !first_ln_done) {   // (the later blocks' first LayerNorm rides on the fc2 GEMM of the block before)
        rc = launch_ln_mod(w.x, b.norm1, mod, mod + H, mod_stride, w.xn, M, L, B, H, st);
        if (rc) return rc;
      }
      GemmParams g = {};
      g.A = w.xn; g.W = b.qkv; g.C = w.qkv; g.M = M; g.N = 3 * H; g.K = H; g.lda = H; g.ldw = H; g.ldc = 3 * H;
      g.L = L; g.B = B; g.rope_cos = m->rope_cos; g.rope_sin = m->rope_sin; g.rope_cols = 2 * H; g.head_dim = 64;
      rc = launch_gemm_bf16(g, EPI_QKV_ROPE, st);
      if (rc) return rc;
      rc = launch_attention(w.qkv, w.att, B, L, c.n_heads, 64, st);
      if (rc) return rc;
      g = GemmParams{};
      g.A = w.att; g.W = b.attn_out; g.C = w.x; g.xin = w.x; g.M = M; g.N = H; g.K = H; g.lda = H; g.ldw = H; g.ldc = H;
      g.L = L; g.B = B; g.gate = mod + 2 * H; g.gate_stride = mod_stride;
      rc = gate_res_then_ln(g, b.norm2, mod + 3 * H, mod + 4 * H);
      if (rc) return rc;
      g = GemmParams{};
      g.A = w.xn; g.W = b.fc1; g.C = w.mid; g.bias = b.fc1_b; g.M = M; g.N = F; g.K = H; g.lda = H; g.ldw = H; g.ldc = F;
      g.L = L; g.B = B;
      rc = launch_gemm_bf16(g, EPI_BIAS_GELU, st);
      if (rc) return rc;
      g = GemmParams{};
      g.A = w.mid; g.W = b.fc2; g.C = w.x; g.xin = w.x; g.bias = b.fc2_b; g.M = M; g.N = H; g.K = F; g.lda = F; g.ldw = F;
      g.ldc = H; g.L = L; g.B = B; g.gate = mod + 5 * H; g.gate_stride = mod_stride;
      const bool last = i + 1 == c.n_blocks;
      const float* nmod = last ? w.mod + (size_t)c.n_blocks * 6 * H : mod + 6 * H;   // next block's (shift, scale) or the final layer's
      rc = gate_res_then_ln(g, last ? m->out_norm : m->blocks[i + 1].norm1, nmod, nmod + H);
      if (rc) return rc;
    }
    xn_ready = true;
    xcur = w.x;
  } else {
    xcur = x_emb;
  }
  if (hidden_out && hidden_out != xcur) {
    const size_t n = (size_t)M * H;
    hipLaunchKernelGGL(copy_f32_kernel, dim3(2048), dim3(256), 0, st, xcur, hidden_out, n);
  }
  // final layer: LN-modulate + Linear(hidden -> vocab)
  const float* fmod = w.mod + (size_t)c.n_blocks * 6 * H;  // shift, scale
  if (!xn_ready) {
    rc = launch_ln_mod(xcur, m->out_norm, fmod, fmod + H, mod_stride, w.xn, M, L, B, H, st);
    if (rc) return rc;
  }
  GemmParams g = {};
  g.A = w.xn; g.W = m->out_lin; g.C = logits; g.bias = m->out_lin_b; g.M = M; g.N = c.vocab; g.K = H; g.lda = H; g.ldw = H;
  g.ldc = c.vocab; g.L = L; g.B = B;
  rc = launch_gemm_bf16(g, EPI_BIAS_F32, st);
  if (rc) return rc;
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}
