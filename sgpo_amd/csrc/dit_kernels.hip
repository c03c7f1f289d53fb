// Non-GEMM kernels of the MDLM denoiser (DiT): conditioning MLP + adaLN tables,
// embedding gather, LayerNorm+modulate, short-sequence attention.
// Reference: models/pretraining/model/mdlm/models/dit.py (line refs per kernel).
#include <stdlib.h>
#include "common.h"
#include "ln_stats.h"
#include "dit.h"
#include "prof.h"

// ---------------------------------------------------------------------------
// RoPE tables  cos/sin[pos][i], i < head_dim/2   (dit.py:81-104: inv_freq = 1/10000^(2i/d))
__global__ void rope_table_kernel(float* cs, float* sn, int max_len, int half) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= max_len * half) return;
  int pos = idx / half, i = idx % half;
  float inv = 1.0f / powf(10000.0f, (float)(2 * i) / (float)(2 * half));
  float a = (float)pos * inv;
  cs[idx] = cosf(a);
  sn[idx] = sinf(a);
}

// ---------------------------------------------------------------------------
// c = silu(W2 silu(W1 emb(sigma) + b1) + b2)   (dit.py:150-189, :385), one block per sigma row.
// Then every adaLN_modulation Linear (dit.py:232, :310): mod[row][j] = Wm[j,:] c + bm[j]
// over the concatenated [n_blocks*6*H + 2*H] rows of all blocks + final layer.
// The reference runs the adaLN Linear under bf16 autocast; here it is fp32 (closer to the
// fp32 CPU oracle; DESIGN.md "precision contract").
__global__ __launch_bounds__(256) void conditioning_kernel(const float* __restrict__ sigma, int n_sigma,
                                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ w2, const float* __restrict__ b2,
                                                           int freq_dim, int cond_dim, float* __restrict__ c_out) {
  extern __shared__ float sh[];  // emb[freq_dim] + h[cond_dim]
  float* emb = sh;
  float* h = sh + freq_dim;
  const int row = blockIdx.x;
  const float s = sigma ? sigma[row] : 0.f;
  const int half = freq_dim / 2;
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    float f = expf(-logf(10000.f) * (float)i / (float)half);
    emb[i] = cosf(s * f);
    emb[half + i] = sinf(s * f);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < cond_dim; j += blockDim.x) {
    float a = b1[j];
    for (int k = 0; k < freq_dim; ++k) a += w1[j * freq_dim + k] * emb[k];
    h[j] = a / (1.f + expf(-a));
  }
  __syncthreads();
  for (int j = threadIdx.x; j < cond_dim; j += blockDim.x) {
    float a = b2[j];
    for (int k = 0; k < cond_dim; ++k) a += w2[j * cond_dim + k] * h[k];
    c_out[row * cond_dim + j] = a / (1.f + expf(-a));
  }
}

__global__ __launch_bounds__(256) void modulation_kernel(const float* __restrict__ c, int cond_dim,
                                                         const float* __restrict__ wm, const float* __restrict__ bm,
                                                         int n_out, float* __restrict__ mod) {
  // grid (ceil(n_out/256), n_sigma)
  extern __shared__ float cs[];
  const int row = blockIdx.y;
  for (int k = threadIdx.x; k < cond_dim; k += blockDim.x) cs[k] = c[row * cond_dim + k];
  __syncthreads();
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_out) return;
  float a = bm[j];
  const float* w = wm + (size_t)j * cond_dim;
  for (int k = 0; k < cond_dim; ++k) a += w[k] * cs[k];
  mod[(size_t)row * n_out + j] = a;
}

// ---------------------------------------------------------------------------
// x0[m][:] = embedding[token[m]][:]   (dit.py:292-299), f32.  One wave per row, float4 lanes.
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ tok, const float* __restrict__ emb,
                                                    float* __restrict__ x, int M, int H, int vocab) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  int64_t t = tok[row];
  if (t < 0) t = 0;
  if (t >= vocab) t = vocab - 1;
  const float4* src = reinterpret_cast<const float4*>(emb + (size_t)t * H);
  float4* dst = reinterpret_cast<float4*>(x + (size_t)row * H);
  for (int i = lane; i < H / 4; i += 64) dst[i] = src[i];
}

// ---------------------------------------------------------------------------
// xn = LN(x) * w * (1 + scale[b]) + shift[b]  -> bf16   (dit.py:126-134, :46-49, :254, :286, :319)
// LN: no bias, eps 1e-5, fp32 statistics.  One wave per row of H = 512 (8 floats per lane),
// two-pass mean/variance in registers.  HBM-bound: 4H read + 2H written per row.
template <int H>
// With `tok` the row is the embedding of token[row] (embed_kernel's gather, dit.py:292-299) and is also written to
// x_out: the first LayerNorm of a forward then costs no read of the residual stream.
__global__ __launch_bounds__(256) void ln_mod_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ shift, const float* __restrict__ scale,
                                                     int mod_stride, bf16_t* __restrict__ out, int M, int L, int B,
                                                     const int64_t* __restrict__ tok, int vocab, float* __restrict__ x_out) {
  constexpr int PER = H / 64;  // floats per lane
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  float v[PER];
  const float* xr = x + (size_t)row * H;
  if (tok) {
    int64_t t = tok[row];
    if (t < 0) t = 0;
    if (t >= vocab) t = vocab - 1;
    xr = x + (size_t)t * H;   // x is the embedding table here
  }
#pragma unroll
  for (int i = 0; i < PER / 4; ++i) {
    float4 t = reinterpret_cast<const float4*>(xr)[i * 64 + lane];
    v[i * 4 + 0] = t.x; v[i * 4 + 1] = t.y; v[i * 4 + 2] = t.z; v[i * 4 + 3] = t.w;
    if (tok) reinterpret_cast<float4*>(x_out + (size_t)row * H)[i * 64 + lane] = t;
  }
  float mean, rstd;
  if constexpr (H == 512) {
    // the summation order of ln_stats.h (= the fused GEMM epilogue): quad i of lane l is quad l % 16 of segment
    // 4 i + l / 16; the eight segment statistics are fetched from the first lane of their DPP row
    float ms[8], qs[8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float m_s, q_s;
      lns_segment(v[i * 4 + 0], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3], m_s, q_s);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ms[i * 4 + r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m_s), 16 * r));
        qs[i * 4 + r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q_s), 16 * r));
      }
    }
    lns_combine8(ms, qs, 1e-5f, mean, rstd);
  } else {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) s += v[i];
    mean = wave_sum(s) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { float d = v[i] - mean; q += d * d; }
    rstd = rsqrtf(wave_sum(q) * (1.0f / H) + 1e-5f);
  }
  int b = row / L;
  if (b > B - 1) b = B - 1;
  const float* sh = shift + (size_t)b * mod_stride;
  const float* sc = scale + (size_t)b * mod_stride;
#pragma unroll
  for (int i = 0; i < PER / 4; ++i) {
    const int col = (i * 64 + lane) * 4;
    float4 ww = *reinterpret_cast<const float4*>(w + col);
    float4 s4 = *reinterpret_cast<const float4*>(sh + col);
    float4 c4 = *reinterpret_cast<const float4*>(sc + col);
    uint2 pk;
    pk.x = pack_bf16x2(lns_modulate(v[i * 4 + 0], mean, rstd, ww.x, c4.x, s4.x), lns_modulate(v[i * 4 + 1], mean, rstd, ww.y, c4.y, s4.y));
    pk.y = pack_bf16x2(lns_modulate(v[i * 4 + 2], mean, rstd, ww.z, c4.z, s4.z), lns_modulate(v[i * 4 + 3], mean, rstd, ww.w, c4.w, s4.w));
    *reinterpret_cast<uint2*>(out + (size_t)row * H + col) = pk;
  }
}

// Generic-H fallback (H % 256 == 0 not required): one wave per row, strided loop.
__global__ __launch_bounds__(256) void ln_mod_generic_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ shift, const float* __restrict__ scale,
                                                             int mod_stride, bf16_t* __restrict__ out, int M, int L, int B, int H) {
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
  int b = row / L;
  if (b > B - 1) b = B - 1;
  const float* sh = shift ? shift + (size_t)b * mod_stride : nullptr;
  const float* sc = scale ? scale + (size_t)b * mod_stride : nullptr;
  for (int i = lane; i < H; i += 64) {
    float o = (xr[i] - mean) * rstd * w[i];
    if (sc) o = o * (1.f + sc[i]) + sh[i];
    out[(size_t)row * H + i] = f2bf(o);
  }
}

int launch_ln_mod(const float* x, const float* w, const float* shift, const float* scale, int mod_stride,
                  bf16_t* out, int M, int L, int B, int H, hipStream_t stream) {
  dim3 grid((M + 3) / 4), block(256);
  ProfScope prof_scope(PK_LN_MOD, stream, (double)M * H * 6.0);  // 4H read + 2H written
  if (H == 512 && shift && scale)
    hipLaunchKernelGGL(ln_mod_kernel<512>, grid, block, 0, stream, x, w, shift, scale, mod_stride, out, M, L, B, nullptr, 0, nullptr);
  else if (H == 1024 && shift && scale)
    hipLaunchKernelGGL(ln_mod_kernel<1024>, grid, block, 0, stream, x, w, shift, scale, mod_stride, out, M, L, B, nullptr, 0, nullptr);
  else
    hipLaunchKernelGGL(ln_mod_generic_kernel, grid, block, 0, stream, x, w, shift, scale, mod_stride, out, M, L, B, H);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

// ---------------------------------------------------------------------------
// Non-causal multi-head attention for short fixed-length sequences (L <= 512), head_dim 64.
// Replaces flash_attn_varlen_qkvpacked_func (dit.py:272-273) for equal-length sequences.
//
// One workgroup (4 waves) per (sequence, head).  K [Lp][64] and V [Lp][64] (both row-major with
// XOR-swizzled 16-B chunks) live in LDS for the whole sequence; V is consumed column-wise through the
// transposing LDS read ds_read_b64_tr_b16 (4 keys x 16 d per 16-lane group), so no V^T image is built.  A wave owns 32-query tiles and walks the
// key tiles with an online softmax.  Products are "swapped" so every per-query quantity
// (max, sum, rescale) is lane-local:
//     S^T[key][query] = K Q^T          A = K rows (LDS),  B = Q (registers)
//     O^T[d][query]  += V^T P^T        A = V^T rows (LDS), B = the S^T accumulator itself,
//                                      converted to bf16 in place (k order of the accumulator).
// qkv: [M][3*Hd] bf16 (q | k | v sections, heads contiguous inside a section), RoPE already applied.
#define ATT_D 64
typedef uint32_t u32x4_att __attribute__((ext_vector_type(4)));
#ifndef ATT_ABL
#define ATT_ABL 0   // diagnostic builds only (tools/dbg/att_bench.hip): 1 no stores, 2 no key loop, 4 no K/V loads
#endif

// One (32-query tile) x (32-key tile) step of the online softmax, shared by both attention kernels so that they give the
// same bits: S^T = K Q^T, running max / sum per query (lane-local + one exchange with lane ^ 32), O^T += V^T P^T.
// k_tile / v_tile: the tile's K rows (chunk c of row r at c ^ ((r >> 1) & 7)) and V rows (c ^ ((r & 3) << 1)) in LDS.
struct AttState { f32x16 o0, o1; float m_run, l_run; };
__device__ __forceinline__ void att_tile(const char* k_tile, const char* v_tile, const bf16x8 (&qf)[4], AttState& a, int kt, int L,
                                         float scale_log2e, int lr, int lh, int v_off0, int v_off1) {
  f32x16 s;
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] = 0.f;
  const char* kr = k_tile + lr * 128;
  const int sw = (lr >> 1) & 7;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    bf16x8 kf = *reinterpret_cast<const bf16x8*>(kr + (((kk * 2 + lh) ^ sw) << 4));
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s, 0, 0, 0);
  }
  // s[r] = S^T[key = kt*32 + (r&3) + 8*(r>>2) + 4*lh][query = lr]  (unscaled)
  if (kt * 32 + 32 > L) {   // only the last key tile can hold padding keys (wave-uniform)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (key >= L) s[r] = -1e30f;
    }
  }
  float tmax = fmaxf(s[0], s[1]);
#pragma unroll
  for (int r = 2; r < 16; r += 2) tmax = __builtin_fmaxf(__builtin_fmaxf(s[r], s[r + 1]), tmax);   // v_max3_f32
  {   // the other 16 keys of the tile are in lane ^ 32: one v_permlane32_swap (VALU) instead of a trip through the LDS crossbar
    const auto hx = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
    tmax = fmaxf(__uint_as_float(hx[0]), __uint_as_float(hx[1])) * scale_log2e;   // scale > 0: max commutes with it
  }
  const float m_new = fmaxf(a.m_run, tmax);
  const float alpha = __builtin_amdgcn_exp2f(a.m_run - m_new);
  float psum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -m_new)); psum += s[r]; }
  {
    const auto hx = __builtin_amdgcn_permlane32_swap(__float_as_uint(psum), __float_as_uint(psum), false, false);
    psum = __uint_as_float(hx[0]) + __uint_as_float(hx[1]);
  }
  a.l_run = a.l_run * alpha + psum;
  a.m_run = m_new;
  // the running maximum settles after the first key tiles; when it moved in no lane, alpha is exactly 1 everywhere and
  // the 32 multiplies are skipped (same bits).  At L = 389 the loop is VALU-bound: 16 v_exp_f32 + these per 8 MFMAs.
  if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { a.o0[r] *= alpha; a.o1[r] *= alpha; }
  }
  // P^T as the B operand of the next product: k-step st uses registers 8st..8st+7; element j of
  // lane-half lh is key 16st + 8(j>>2) + 4 lh + (j&3) of this tile.
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    u32x4_att pw;   // one v_cvt_pk_bf16_f32 per pair
    pw.x = pack_bf16x2(s[st * 8 + 0], s[st * 8 + 1]); pw.y = pack_bf16x2(s[st * 8 + 2], s[st * 8 + 3]);
    pw.z = pack_bf16x2(s[st * 8 + 4], s[st * 8 + 5]); pw.w = pack_bf16x2(s[st * 8 + 6], s[st * 8 + 7]);
    const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
    // A operand: V^T[d = lr (+32)][keys kt*32 + st*16 + 4 lh + {0..3, 8..11}] by transposed reads of V
    const char* vb = v_tile + (st * 16) * 128;
    typedef short s16x4_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_t;
    const s16x4_t va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vb + v_off0));
    const s16x4_t vc = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vb + v_off0 + 8 * 128));
    const s16x4_t vd = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vb + v_off1));
    const s16x4_t ve = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vb + v_off1 + 8 * 128));
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    const s16x8_t v0s = __builtin_shufflevector(va, vc, 0, 1, 2, 3, 4, 5, 6, 7);
    const s16x8_t v1s = __builtin_shufflevector(vd, ve, 0, 1, 2, 3, 4, 5, 6, 7);
    const bf16x8 v0 = __builtin_bit_cast(bf16x8, v0s), v1 = __builtin_bit_cast(bf16x8, v1s);
    a.o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pf, a.o0, 0, 0, 0);
    a.o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pf, a.o1, 0, 0, 0);
  }
}

// ---- O[query][d] = O^T / l ; lane owns query q, rows d = (r&3)+8(r>>2)+4lh (+32), i.e. the 8-byte
// halves of 16-byte chunk g (and 4+g) of its output row, the other half being in lane^32.  Stored like that,
// a wave's 16 store instructions visit 512 cache lines and the CU's address unit (about one line per cycle)
// costs the kernel a third of its time.  One v_permlane32_swap per register instead hands lane half 0 the
// even chunks and half 1 the odd ones, whole: 4 instructions of 16 bytes per lane.
__device__ __forceinline__ void att_store(const AttState& a, bf16_t* orow_base, bool valid, int lh) {
  const float inv = 1.0f / a.l_run;
  uint32_t w0[8], w1[8];   // w[2g], w[2g+1]: this lane's half of chunk g
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    w0[2 * g] = pack_bf16x2(a.o0[g * 4 + 0] * inv, a.o0[g * 4 + 1] * inv);
    w0[2 * g + 1] = pack_bf16x2(a.o0[g * 4 + 2] * inv, a.o0[g * 4 + 3] * inv);
    w1[2 * g] = pack_bf16x2(a.o1[g * 4 + 0] * inv, a.o1[g * 4 + 1] * inv);
    w1[2 * g + 1] = pack_bf16x2(a.o1[g * 4 + 2] * inv, a.o1[g * 4 + 3] * inv);
  }
  // swap(X = half of chunk 2k, Y = half of chunk 2k+1): lanes 32..63 of X <-> lanes 0..31 of Y.  Afterwards the
  // lower lanes hold chunk 2k as (X, Y) and the upper lanes chunk 2k+1 as (X, Y).
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      auto r0 = __builtin_amdgcn_permlane32_swap(w0[4 * k + j], w0[4 * k + 2 + j], false, false);
      w0[4 * k + j] = r0[0]; w0[4 * k + 2 + j] = r0[1];
      auto r1 = __builtin_amdgcn_permlane32_swap(w1[4 * k + j], w1[4 * k + 2 + j], false, false);
      w1[4 * k + j] = r1[0]; w1[4 * k + 2 + j] = r1[1];
    }
  if (valid && (!(ATT_ABL & 1) || a.l_run == 12345.f)) {
    bf16_t* orow = orow_base + lh * 8;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      *reinterpret_cast<uint4*>(orow + 16 * k) = make_uint4(w0[4 * k], w0[4 * k + 1], w0[4 * k + 2], w0[4 * k + 3]);
      *reinterpret_cast<uint4*>(orow + 32 + 16 * k) = make_uint4(w1[4 * k], w1[4 * k + 1], w1[4 * k + 2], w1[4 * k + 3]);
    }
  }
}

__device__ __forceinline__ void att_block_to_bh(int n_heads, int& b, int& h) {
  // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8).  The heads of one sequence read
  // neighbouring 128-B pieces of the same qkv rows, so they are given ids 8 apart: same XCD (one L2),
  // dispatched back to back.  Sequences beyond the last multiple of 8 use the plain order.
  const int per = 8 * n_heads, grp = blockIdx.x / per, r = blockIdx.x - grp * per;
  const int n_seq = gridDim.x / n_heads;
  if ((grp + 1) * 8 <= n_seq) { b = grp * 8 + (r & 7); h = r >> 3; }
  else { b = blockIdx.x / n_heads; h = blockIdx.x % n_heads; }
}

// NW waves per workgroup: 4 for short sequences (5 workgroups per CU), 8 once the K/V images leave room for two
// workgroups per CU (L > 160: twice the waves in flight behind the same LDS bytes), 16 when only one fits (L > 320)
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 5 : NW == 8 ? 2 : 1) void attention_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                         int L, int Lp, int n_heads, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int hid = n_heads * ATT_D;
  const int ld = 3 * hid;
  int b, h;
  att_block_to_bh(n_heads, b, h);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  char* k_lds = smem;              // Lp * 128 B, chunk c of row r at c ^ ((r >> 1) & 7)
  char* v_lds = smem + Lp * 128;   // Lp * 128 B, chunk c of row r at c ^ ((r & 3) << 1)
  const bf16_t* base = qkv + (size_t)b * L * ld + h * ATT_D;

  // Q fragments of this wave's first query tile are requested before the K/V staging so that both latencies overlap.
  // B operand: lane holds Q[query = qt*32+lr][d = kk*16 + 8*lh + j]
  bf16x8 qf[4];
  {
    int qrow = wave * 32 + lr;
    if (qrow > L - 1) qrow = L - 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      qf[kk] = *reinterpret_cast<const bf16x8*>(base + (size_t)qrow * ld + kk * 16 + lh * 8);
  }
  // ---- stage K (swizzled rows) and V; rows >= L are zero.  Four rounds of loads are requested before the first
  // LDS write, so a 128-row sequence pays one memory latency instead of four.
  for (int i0 = tid; i0 < Lp * 8; i0 += 64 * NW * 4) {
    uint4 kv[4], vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = i0 + u * 64 * NW, row = idx >> 3, c = idx & 7;
      kv[u] = make_uint4(0, 0, 0, 0);
      vv[u] = make_uint4(0, 0, 0, 0);
      if (row < L && !(ATT_ABL & 4)) {   // covers idx >= Lp * 8 as well (L <= Lp)
        kv[u] = *reinterpret_cast<const uint4*>(base + (size_t)row * ld + hid + c * 8);
        vv[u] = *reinterpret_cast<const uint4*>(base + (size_t)row * ld + 2 * hid + c * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = i0 + u * 64 * NW, row = idx >> 3, c = idx & 7;
      if (idx < Lp * 8) {
        *reinterpret_cast<uint4*>(k_lds + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = kv[u];
        *reinterpret_cast<uint4*>(v_lds + row * 128 + ((c ^ ((row & 3) << 1)) << 4)) = vv[u];
      }
    }
  }
  __syncthreads();
  // transposed-read addressing (per 16-lane group: lane 4q+p supplies row q, columns 4p..4p+3 of a 4-key x
  // 16-d block and lane i receives column i): this lane's d is lr, its keys are 4 lh + {0..3} (+8)
  const int tq = (lane >> 2) & 3, tp = lane & 3, dblk = lr >> 4;
  const int v_off0 = (lh * 4 + tq) * 128 + (((dblk * 2 + (tp >> 1)) ^ (tq << 1)) << 4) + (tp & 1) * 8;        // d in [0, 32)
  const int v_off1 = (lh * 4 + tq) * 128 + (((4 + dblk * 2 + (tp >> 1)) ^ (tq << 1)) << 4) + (tp & 1) * 8;    // d in [32, 64)

  const int n_tiles = Lp / 32;
  for (int qt = wave; qt < n_tiles; qt += NW) {
    if (qt != wave) {
      int qrow = qt * 32 + lr;
      if (qrow > L - 1) qrow = L - 1;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        qf[kk] = *reinterpret_cast<const bf16x8*>(base + (size_t)qrow * ld + kk * 16 + lh * 8);
    }
    AttState a;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a.o0[r] = 0.f; a.o1[r] = 0.f; }
    a.m_run = -1e30f; a.l_run = 0.f;
    for (int kt = 0; kt < ((ATT_ABL & 2) ? 0 : n_tiles); ++kt)
      att_tile(k_lds + kt * 32 * 128, v_lds + kt * 32 * 128, qf, a, kt, L, scale_log2e, lr, lh, v_off0, v_off1);
    const int q = qt * 32 + lr;
    att_store(a, out + ((size_t)b * L + q) * hid + h * ATT_D, q < L, lh);
  }
}

// Long sequences (L > 320, where the K/V images of one head leave room for a single workgroup per CU and nothing
// overlaps its staging with its key loop): K/V tiles of 32 keys stream through a 4-slot LDS ring (LDS-DMA, one 1-KiB
// piece per wave and stage, counted waits, one barrier per key tile), 32 KiB per workgroup, so two 8-wave workgroups
// share a CU.  A wave owns query tiles wave, wave + 8; each round of query tiles streams the keys again (from L2).
// Same tile arithmetic (att_tile) as the resident kernel: same bits.  Rows beyond L are read from row L - 1 (finite
// values; their probabilities are exactly 0).
#define ATT_NSTG 4
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void attention_stream_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int L,
                                                                                    int Lp, int n_heads, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // ATT_NSTG x (K tile 4 KiB | V tile 4 KiB)
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int hid = n_heads * ATT_D;
  const int ld = 3 * hid;
  int b, h;
  att_block_to_bh(n_heads, b, h);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const bf16_t* base = qkv + (size_t)b * L * ld + h * ATT_D;
  const int tq = (lane >> 2) & 3, tp = lane & 3, dblk = lr >> 4;
  const int v_off0 = (lh * 4 + tq) * 128 + (((dblk * 2 + (tp >> 1)) ^ (tq << 1)) << 4) + (tp & 1) * 8;
  const int v_off1 = (lh * 4 + tq) * 128 + (((4 + dblk * 2 + (tp >> 1)) ^ (tq << 1)) << 4) + (tp & 1) * 8;
  const int n_tiles = Lp / 32;
  // this wave's piece of every stage: waves 0-3 rows 8w..8w+7 of the K tile, waves 4-7 of the V tile (waves 8+ stage nothing)
  const bool stager = wave < 8;
  const bool is_v = wave >= 4;
  const int prow = (wave & 3) * 8 + (lane >> 3), pc = lane & 7;
  const int src_c = pc ^ (is_v ? ((prow & 3) << 1) : ((prow >> 1) & 7));   // the swizzle is applied on the source side
  const bf16_t* src0 = base + (is_v ? 2 * hid : hid) + src_c * 8;
  const int dst_off = (is_v ? 4096 : 0) + (wave & 3) * 1024;
  auto stage = [&](int kt) {
    if (!stager) return;
    int row = kt * 32 + prow;
    if (row > L - 1) row = L - 1;
    __builtin_amdgcn_global_load_lds((gptr_t)(src0 + (size_t)row * ld), (lptr_t)(smem + (kt % ATT_NSTG) * 8192 + dst_off), 16, 0, 0);
  };
  for (int q0 = 0; q0 < n_tiles; q0 += NW) {
    const int qt = q0 + wave;
    const bool active = qt < n_tiles;    // wave-uniform; idle waves still stage and meet the barriers
    bf16x8 qf[4];
    {
      int qrow = qt * 32 + lr;
      if (qrow > L - 1) qrow = L - 1;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        qf[kk] = *reinterpret_cast<const bf16x8*>(base + (size_t)qrow * ld + kk * 16 + lh * 8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // Q fragments (and the stores of the round before) are out of the queue
#pragma unroll
    for (int s0 = 0; s0 < ATT_NSTG - 1; ++s0)
      if (s0 < n_tiles) stage(s0);
    AttState a;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a.o0[r] = 0.f; a.o1[r] = 0.f; }
    a.m_run = -1e30f; a.l_run = 0.f;
    for (int kt = 0; kt < n_tiles; ++kt) {
      const int ahead = min(ATT_NSTG - 2, n_tiles - 1 - kt);   // younger stages that may still be in flight
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // every wave's piece of tile kt is visible; every read of tile kt - 1 is done
      asm volatile("" ::: "memory");
      if (kt + ATT_NSTG - 1 < n_tiles) stage(kt + ATT_NSTG - 1);
      const char* slot = smem + (kt % ATT_NSTG) * 8192;
      if (active && !(ATT_ABL & 2)) att_tile(slot, slot + 4096, qf, a, kt, L, scale_log2e, lr, lh, v_off0, v_off1);
    }
    if (active) {
      const int q = qt * 32 + lr;
      att_store(a, out + ((size_t)b * L + q) * hid + h * ATT_D, q < L, lh);
    }
    __syncthreads();   // the next round's first stages reuse the slots the last tiles were read from
  }
}

int launch_attention(const bf16_t* qkv, bf16_t* out, int B, int L, int n_heads, int head_dim, hipStream_t stream) {
  SGPO_REQUIRE(head_dim == ATT_D, "attention: head_dim %d unsupported (64 only)", head_dim);
  const int Lp = (L + 31) / 32 * 32;
  const size_t lds = (size_t)Lp * 256;   // K and V images
  static int stream_mode = -1;
  if (stream_mode < 0) { const char* e = getenv("SGPO_ATT_STREAM"); stream_mode = (e && e[0] == '0') ? 0 : (e && e[0] == '2') ? 2 : (e && e[0] == '3') ? 3 : 1; }   // A/B measurements
  // the streaming kernel has no length limit of its own; only the resident kernels need both images in LDS
  SGPO_REQUIRE(stream_mode != 0 || lds <= 160 * 1024, "attention: sequence length %d exceeds the LDS-resident limit", L);
  static size_t attr_lds = 0;
  if (lds > attr_lds && lds <= 160 * 1024) {
    SGPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<4>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SGPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<8>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SGPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<16>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_lds = lds;
  }
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)head_dim);
  ProfScope prof_scope(PK_ATTENTION, stream, 4.0 * B * n_heads * (double)L * L * head_dim);
  if (lds > 80 * 1024 && stream_mode == 2)
    hipLaunchKernelGGL(attention_stream_kernel<16>, dim3(B * n_heads), dim3(1024), ATT_NSTG * 8192, stream, qkv, out, L, Lp, n_heads, scale_log2e);
  else if ((lds > 80 * 1024 && stream_mode) || (lds > 40 * 1024 && stream_mode == 3))   // the resident images would leave one workgroup per CU: stream the keys instead
    hipLaunchKernelGGL(attention_stream_kernel<8>, dim3(B * n_heads), dim3(512), ATT_NSTG * 8192, stream, qkv, out, L, Lp, n_heads, scale_log2e);
  else if (lds > 80 * 1024)   // one workgroup per CU: 16 waves cover up to 16 query tiles in one round
    hipLaunchKernelGGL(attention_kernel<16>, dim3(B * n_heads), dim3(1024), lds, stream, qkv, out, L, Lp, n_heads, scale_log2e);
  else if (lds > 40 * 1024)
    hipLaunchKernelGGL(attention_kernel<8>, dim3(B * n_heads), dim3(512), lds, stream, qkv, out, L, Lp, n_heads, scale_log2e);
  else
    hipLaunchKernelGGL(attention_kernel<4>, dim3(B * n_heads), dim3(256), lds, stream, qkv, out, L, Lp, n_heads, scale_log2e);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

int launch_embed(const int64_t* tok, const float* emb, float* x, int M, int H, int vocab, hipStream_t stream) {
  ProfScope prof_scope(PK_EMBED, stream, (double)M * (H * 4.0 + 8.0));   // fp32 row written, one int64 token read (the table stays in cache)
  hipLaunchKernelGGL(embed_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, tok, emb, x, M, H, vocab);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

// embedding gather + the first LayerNorm-modulate in one pass (H = 512 / 1024); SGPO_ERR_UNSUPPORTED otherwise
int launch_embed_ln_mod(const int64_t* tok, const float* emb, int vocab, float* x, const float* w, const float* shift,
                        const float* scale, int mod_stride, bf16_t* out, int M, int L, int B, int H, hipStream_t stream) {
  if ((H != 512 && H != 1024) || !shift || !scale) return SGPO_ERR_UNSUPPORTED;
  dim3 grid((M + 3) / 4), block(256);
  ProfScope prof_scope(PK_EMBED, stream, (double)M * (H * 6.0 + 8.0));   // fp32 row + bf16 row written, one token read
  if (H == 512)
    hipLaunchKernelGGL(ln_mod_kernel<512>, grid, block, 0, stream, emb, w, shift, scale, mod_stride, out, M, L, B, tok, vocab, x);
  else
    hipLaunchKernelGGL(ln_mod_kernel<1024>, grid, block, 0, stream, emb, w, shift, scale, mod_stride, out, M, L, B, tok, vocab, x);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

int launch_rope_table(float* cs, float* sn, int max_len, int half, hipStream_t stream) {
  const int n = max_len * half;
  hipLaunchKernelGGL(rope_table_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, cs, sn, max_len, half);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

int launch_conditioning(const float* sigma, int n_sigma, const float* w1, const float* b1, const float* w2,
                        const float* b2, int freq_dim, int cond_dim, float* c_out, const float* wm,
                        const float* bm, int n_out, float* mod, hipStream_t stream) {
  hipLaunchKernelGGL(conditioning_kernel, dim3(n_sigma), dim3(256), (freq_dim + cond_dim) * sizeof(float), stream,
                     sigma, n_sigma, w1, b1, w2, b2, freq_dim, cond_dim, c_out);
  hipLaunchKernelGGL(modulation_kernel, dim3((n_out + 255) / 256, n_sigma), dim3(256), cond_dim * sizeof(float),
                     stream, c_out, cond_dim, wm, bm, n_out, mod);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}
