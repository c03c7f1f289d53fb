// Row f-3: value-function ensemble training (training/train_classifier.py:7-64 -- Adam on the MSE of
// MLPModel, problem/protein.py:58-121), all ensemble members in lockstep, five launches per optimiser step.
//   tr_fwd0   grid (chunk groups, members): two 64-wide input chunks (a residue's embedding each, or the time
//             embedding) times their slices of mlp.0.weight, summed in the MFMA accumulators -> partial pre-activations
//   tr_reduce grid over (member, row, unit): ordered sum of the partials + bias
//   tr_mid    grid (members): the four 64x64 layers + head forward, MSE, and the
//             backward pass down to d(pre-activation of mlp.0); activations live in LDS
//   tr_bwd0   grid (chunks, members): gradient of the mlp.0.weight slice and Adam on it (96 % of the parameters: that
//             gradient never leaves the chip), gradient of the chunk's input and through the embedder (per-chunk
//             partial sums per token) or the time embedder
//   tr_adam   grid over the other parameters: fixed-shape sum of the embedder partials, torch.optim.Adam's update
// fp32 throughout (the 64x64x64 products on the matrix cores), every reduction in a fixed order (no atomics): a run
// is reproducible bit for bit.
#include "common.h"
#include <new>

namespace {

constexpr int H = 64;          // hidden_dim (problem/protein.py: 64)
constexpr int TB = 64;         // largest batch of one step (configs/problem/protein_classifier_discrete.yaml: 64)
constexpr int PAD = H + 1;

struct TrDev {
  float* params;   // [E][P]
  float* grads;    // [E][P]
  float* m;        // [E][P]
  float* v;        // [E][P]
  float* part_h;   // [E][C][TB][H]
  float* dh;       // [E][TB][H]
  float* fbuf;     // [E][TB][H]   Fourier features of t
  float* part_e;   // [E][n_read][32][H]   rows 0..A-1: d embedder.weight[:, s], row 31: d embedder.bias
  const int* read_pos;
  int E, P, C, n_read, A, D, time_cond;
  // offsets into one member's blob (MLP_ORDER of sgpo_amd/runtime.py)
  int o_tw, o_t1w, o_t1b, o_ew, o_eb, o_w0, o_b0, o_w1, o_b1, o_w2, o_b2, o_c0, o_c0b, o_c2, o_c2b;
};


// ---- 64 x 64 x 64 products on the matrix cores, fp32 (v_mfma_f32_32x32x2_f32) ----
// A workgroup of 4 waves forms C = A B; wave w owns the 32 x 32 quadrant (w >> 1, w & 1).  a_at(m, k) and b_at(k, n)
// read LDS (one element each per k pair: 64 reads per lane and product, against 1088 for the FMA form that bounded
// these kernels); `init` is the accumulator's starting value (the bias of the lane's column).  The summation order
// is the MFMA's, fixed: a run stays reproducible bit for bit.
struct Quad {
  int row0, col;   // acc[r] is element (row0 + (r >> 2) * 8 + (r & 3), col)
};
__device__ __forceinline__ Quad quad_of_lane() {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  return Quad{(w >> 1) * 32 + (lane >> 5) * 4, (w & 1) * 32 + (lane & 31)};
}
template <class FA, class FB>
__device__ __forceinline__ f32x16 mfma64(FA a_at, FB b_at, float init) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int m = (w >> 1) * 32 + (lane & 31), n = (w & 1) * 32 + (lane & 31), kh = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = init;
#pragma unroll 8
  for (int k = 0; k < H; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_at(m, k + kh), b_at(k + kh, n), acc, 0, 0, 0);
  return acc;
}
template <class FA, class FB>
__device__ __forceinline__ void mfma64_acc(FA a_at, FB b_at, f32x16& acc) {   // acc += A B
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int m = (w >> 1) * 32 + (lane & 31), n = (w & 1) * 32 + (lane & 31), kh = lane >> 5;
#pragma unroll 8
  for (int k = 0; k < H; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_at(m, k + kh), b_at(k + kh, n), acc, 0, 0, 0);
}
template <class F>
__device__ __forceinline__ void quad_store(const f32x16& acc, F put) {   // put(row, col, value)
  const Quad q = quad_of_lane();
#pragma unroll
  for (int r = 0; r < 16; ++r) put(q.row0 + (r >> 2) * 8 + (r & 3), q.col, acc[r]);
}

// in[b][k] of chunk c: embedding of the token at read position c, or the time embedding (c == n_read)
template <int P1>
__device__ void chunk_input(const TrDev& d, const float* p, int e, int c, const int64_t* x, const float* t, int B, int L,
                            float (*in)[P1], float (*tmp)[P1], bool write_f) {
  const int tid = threadIdx.x;
  if (c < d.n_read) {
    const float* ew = p + d.o_ew;
    const float* eb = p + d.o_eb;
    const int pos = d.read_pos[c];
    for (int idx = tid; idx < TB * H; idx += 256) {
      const int b = idx >> 6, k = idx & 63;
      float v = 0.f;
      if (b < B) {
        int tok = (int)x[((size_t)e * B + b) * L + pos];
        tok = tok < 0 ? 0 : (tok >= d.A ? d.A - 1 : tok);
        v = ew[k * d.A + tok] + eb[k];
      }
      in[b][k] = v;
    }
  } else {
    const float* tw = p + d.o_tw;
    if (write_f) {
      // GaussianFourierProjection (problem/protein.py:54-56): x*W*2*pi, cat(sin, cos).  One sincosf per argument
      // (sin lands in column j, cos in column j + 32): the arguments are large (t W 2 pi up to ~1e4), so the range
      // reduction is the cost, and a sinf and a cosf per argument would do it twice.
      for (int idx = tid; idx < TB * (H / 2); idx += 256) {
        const int b = idx >> 5, j = idx & 31;
        float sn = 0.f, cs = 0.f;
        if (b < B) {
          const float arg = t[(size_t)e * B + b] * tw[j] * 2.0f * 3.14159274101257324f;
          sincosf(arg, &sn, &cs);
          d.fbuf[((size_t)e * TB + b) * H + j] = sn;
          d.fbuf[((size_t)e * TB + b) * H + j + 32] = cs;
        }
        tmp[b][j] = sn;
        tmp[b][j + 32] = cs;
      }
    } else {
      for (int idx = tid; idx < TB * H; idx += 256) {
        const int b = idx >> 6, j = idx & 63;
        tmp[b][j] = b < B ? d.fbuf[((size_t)e * TB + b) * H + j] : 0.f;   // the backward pass reads what the forward pass stored
      }
    }
    // Linear(64, 64) of the time embedder.  Its weight goes through `in` (row j read by lane j) before `in` takes
    // the result: read straight from memory, lane j walking row j is 64 cache lines per load instruction, and the one
    // workgroup per member that owns this chunk would set the kernel's duration.
    const float* w = p + d.o_t1w;
    const float* bb = p + d.o_t1b;
    for (int idx = tid; idx < H * H; idx += 256) in[idx >> 6][idx & 63] = w[idx];
    __syncthreads();
    // out[b][j] = bias[j] + sum_k f[b][k] W[j][k]
    const f32x16 acc = mfma64([&](int m, int k) { return tmp[m][k]; }, [&](int k, int n) { return in[n][k]; }, bb[quad_of_lane().col]);
    __syncthreads();
    quad_store(acc, [&](int b, int j, float v) { in[b][j] = b < B ? v : 0.f; });
  }
}

constexpr int FWD_CHUNKS = 2;   // input chunks per tr_fwd0 workgroup, summed in the MFMA accumulators (1: 29 + 8 us for forward + reduce, two rounds of workgroups; 4: 38 + 5, the chain per workgroup is too long)

__global__ __launch_bounds__(256) void tr_fwd0(TrDev d, const int64_t* __restrict__ x, const float* __restrict__ t, int B, int L) {
  __shared__ float in[TB][PAD], ws[H][PAD];
  const int n_groups = gridDim.x;
  const int grp = (blockIdx.x + n_groups - 1) % n_groups, e = blockIdx.y, tid = threadIdx.x;   // the group with the (heavier) time chunk, the last one, is dispatched first
  const float* p = d.params + (size_t)e * d.P;
  const int c0 = grp * FWD_CHUNKS, c1 = c0 + FWD_CHUNKS < d.C ? c0 + FWD_CHUNKS : d.C;
  // a weight slice is requested before the input chunk is gathered (token -> embedding is two dependent loads), the
  // next chunk's before this chunk's product: one memory latency per chunk instead of three
  float wreg[H * H / 256];
  auto request_w = [&](int c) {
    const float* w0 = p + d.o_w0 + (size_t)c * H;
#pragma unroll
    for (int it = 0; it < H * H / 256; ++it) { const int idx = tid + it * 256; wreg[it] = w0[(size_t)(idx >> 6) * d.D + (idx & 63)]; }
  };
  request_w(c0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int c = c0; c < c1; ++c) {
    chunk_input(d, p, e, c, x, t, B, L, in, ws, true);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < H * H / 256; ++it) { const int idx = tid + it * 256; ws[idx >> 6][idx & 63] = wreg[it]; }
    if (c + 1 < c1) request_w(c + 1);
    __syncthreads();
    // partial pre-activation: out[b][o] += sum_k in[b][k] W0[o][c * 64 + k]
    mfma64_acc([&](int m, int k) { return in[m][k]; }, [&](int k, int n) { return ws[n][k]; }, acc);
    __syncthreads();
  }
  float* out = d.part_h + (((size_t)e * n_groups + grp) * TB) * H;
  quad_store(acc, [&](int b, int o, float v) { out[b * H + o] = v; });
}

// ordered sum of the chunk-group partials + bias -> pre-activation of mlp.0 (written over group 0's slot)
__global__ void tr_reduce(TrDev d, int B, int n_groups) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (e, b, o)
  if (i >= d.E * TB * H) return;
  const int e = i / (TB * H), bo = i % (TB * H), b = bo >> 6, o = bo & 63;
  if (b >= B) return;
  float* ph = d.part_h + (size_t)e * n_groups * TB * H;
  float s = d.params[(size_t)e * d.P + d.o_b0 + o];
  int c = 0;
  for (; c + 8 <= n_groups; c += 8) {   // eight loads in flight, adds in group order
    float v8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v8[u] = ph[(size_t)(c + u) * TB * H + bo];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v8[u];
  }
  for (; c < n_groups; ++c) s += ph[(size_t)c * TB * H + bo];
  ph[bo] = s;
}

// ---- block-level 64-wide layer helpers (256 threads, rows >= B hold zeros) ----
// out[b][o] = act(bias[o] + sum_k in[b][k] W[o][k])
__device__ void lin_fwd(float (*out)[PAD], float (*in)[PAD], float (*wb)[PAD], const float* bias, bool relu, int B) {
  const f32x16 acc = mfma64([&](int m, int k) { return in[m][k]; }, [&](int k, int n) { return wb[n][k]; }, bias[quad_of_lane().col]);
  quad_store(acc, [&](int b, int o, float v) {
    if (relu) v = v > 0.f ? v : 0.f;
    out[b][o] = b < B ? v : 0.f;
  });
  __syncthreads();
}
// dW[o][k] = sum_b dz[b][o] a[b][k];  db[o] = sum_b dz[b][o]   (rows >= B of dz are zero)
__device__ void grad_w(float* dW, float* db, float (*dz)[PAD], float (*a)[PAD], int B) {
  const f32x16 acc = mfma64([&](int m, int k) { return dz[k][m]; }, [&](int k, int n) { return a[k][n]; }, 0.f);
  quad_store(acc, [&](int o, int k, float v) { dW[o * H + k] = v; });
  if (threadIdx.x < H) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dz[b][threadIdx.x];
    db[threadIdx.x] = s;
  }
}
// da[b][k] = (sum_o dz[b][o] W[o][k]) * (mask ? act[b][k] > 0 : 1);  da may be dz or act (the products are in
// registers before the barrier that precedes the stores)
__device__ void grad_in(float (*da)[PAD], float (*dz)[PAD], float (*wb)[PAD], float (*act)[PAD], bool relu_mask, int B) {
  const f32x16 acc = mfma64([&](int m, int k) { return dz[m][k]; }, [&](int k, int n) { return wb[k][n]; }, 0.f);
  __syncthreads();
  quad_store(acc, [&](int b, int k, float v) {
    if (relu_mask && !(act[b][k] > 0.f)) v = 0.f;
    da[b][k] = b < B ? v : 0.f;
  });
  __syncthreads();
}

__global__ __launch_bounds__(256) void tr_mid(TrDev d, const float* __restrict__ y, int B, int n_groups, float* __restrict__ loss_out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float (*a0)[PAD] = reinterpret_cast<float (*)[PAD]>(smem_raw);
  float (*a1)[PAD] = a0 + TB;
  float (*a2)[PAD] = a1 + TB;
  float (*a3)[PAD] = a2 + TB;
  float (*w1)[PAD] = a3 + TB;        // mlp.2, mlp.4, cls_head.0: loaded once, used by the forward and the backward pass
  float (*w2)[PAD] = w1 + H;
  float (*w3)[PAD] = w2 + H;
  float (*g0)[PAD] = w3 + H;
  float (*g1)[PAD] = g0 + TB;
  float* pred = reinterpret_cast<float*>(g1 + TB);   // [TB]
  float* dpred = pred + TB;                          // [TB]
  const int e = blockIdx.x, tid = threadIdx.x;
  const float* p = d.params + (size_t)e * d.P;
  float* g = d.grads + (size_t)e * d.P;
  // the three 64 x 64 weights in one go (one memory latency instead of one per use), then
  // pre-activation of mlp.0 (tr_reduce left it in chunk 0's slot) -> ReLU
  for (int idx = tid; idx < H * H; idx += 256) {
    w1[idx >> 6][idx & 63] = p[d.o_w1 + idx];
    w2[idx >> 6][idx & 63] = p[d.o_w2 + idx];
    w3[idx >> 6][idx & 63] = p[d.o_c0 + idx];
  }
  {
    const float* ph = d.part_h + (size_t)e * n_groups * TB * H;
    for (int idx = tid; idx < TB * H; idx += 256) {
      const int b = idx >> 6, o = idx & 63;
      float s = 0.f;
      if (b < B) {
        s = ph[idx];
        s = s > 0.f ? s : 0.f;
      }
      a0[b][o] = s;
    }
    __syncthreads();
  }
  lin_fwd(a1, a0, w1, p + d.o_b1, true, B);
  lin_fwd(a2, a1, w2, p + d.o_b2, false, B);
  lin_fwd(a3, a2, w3, p + d.o_c0b, true, B);
  // head + MSE (criterion = MSELoss(): mean over the batch)
  const float* c2 = p + d.o_c2;
  if (tid < TB) {
    float s = 0.f, dp = 0.f;
    if (tid < B) {
      s = p[d.o_c2b];
      for (int k = 0; k < H; ++k) s = fmaf(a3[tid][k], c2[k], s);
      dp = 2.0f * (s - y[(size_t)e * B + tid]) / (float)B;
    }
    pred[tid] = s;
    dpred[tid] = dp;
  }
  __syncthreads();
  if (tid == 0) {
    float l = 0.f;
    for (int b = 0; b < B; ++b) { const float r = pred[b] - y[(size_t)e * B + b]; l += r * r; }
    loss_out[e] = l / (float)B;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dpred[b];
    g[d.o_c2b] = s;
  }
  if (tid < H) {   // d cls_head.2.weight
    float s = 0.f;
    for (int b = 0; b < B; ++b) s = fmaf(dpred[b], a3[b][tid], s);
    g[d.o_c2 + tid] = s;
  }
  // dz3[b][k] = dpred[b] * c2[k] * (a3 > 0)
  for (int idx = tid; idx < TB * H; idx += 256) {
    const int b = idx >> 6, k = idx & 63;
    g0[b][k] = (b < B && a3[b][k] > 0.f) ? dpred[b] * c2[k] : 0.f;
  }
  __syncthreads();
  grad_w(g + d.o_c0, g + d.o_c0b, g0, a2, B);          // cls_head.0
  grad_in(g1, g0, w3, a2, false, B);                   // d a2 (mlp.4 has no ReLU after it)
  grad_w(g + d.o_w2, g + d.o_b2, g1, a1, B);           // mlp.4
  grad_in(g0, g1, w2, a1, true, B);                    // d z of mlp.2
  grad_w(g + d.o_w1, g + d.o_b1, g0, a0, B);           // mlp.2
  grad_in(g1, g0, w1, a0, true, B);                    // d pre-activation of mlp.0
  float* dh = d.dh + (size_t)e * TB * H;
  for (int idx = tid; idx < TB * H; idx += 256) dh[idx] = g1[idx >> 6][idx & 63];
  if (tid < H) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += g1[b][tid];
    g[d.o_b0 + tid] = s;
  }
}

// torch.optim.Adam (single-tensor path): exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad,
// value=1 - beta2); denom = (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps); param.addcdiv_(exp_avg, denom, value=-step_size)
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float gr, float lr_over_bc1, float bc2_sqrt,
                                            float beta1, float beta2, float eps) {
  m = m + (gr - m) * (1.0f - beta1);
  v = v * beta2 + (1.0f - beta2) * gr * gr;
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - lr_over_bc1 * (m / denom);
}
__global__ __launch_bounds__(256) void tr_bwd0(TrDev d, const int64_t* __restrict__ x, const float* __restrict__ t, int B, int L,
                                               float lr_over_bc1, float bc2_sqrt, float beta1, float beta2, float eps) {
  __shared__ float in[TB][PAD], ws[H][PAD], dh[TB][PAD];   // 50 KB: three workgroups per CU
  __shared__ int toks[TB];
  const int c = (blockIdx.x + d.C - 1) % d.C, e = blockIdx.y, tid = threadIdx.x;   // the (heavier) time chunk, the last one, is dispatched first
  const float* p = d.params + (size_t)e * d.P;
  float* g = d.grads + (size_t)e * d.P;
  const float* w0 = p + d.o_w0 + (size_t)c * H;
  const float* dhg = d.dh + (size_t)e * TB * H;
  float wreg[H * H / 256];   // requested before the gather of the input chunk, like tr_fwd0
#pragma unroll
  for (int it = 0; it < H * H / 256; ++it) {
    const int idx = tid + it * 256;
    wreg[it] = w0[(size_t)(idx >> 6) * d.D + (idx & 63)];
    dh[idx >> 6][idx & 63] = dhg[idx];
  }
  chunk_input(d, p, e, c, x, t, B, L, in, ws, false);   // for the time chunk ws holds the Fourier features meanwhile
  __syncthreads();
#pragma unroll
  for (int it = 0; it < H * H / 256; ++it) { const int idx = tid + it * 256; ws[idx >> 6][idx & 63] = wreg[it]; }
  if (tid < TB && c < d.n_read) {
    int tok = 0;
    if (tid < B) {
      tok = (int)x[((size_t)e * B + tid) * L + d.read_pos[c]];
      tok = tok < 0 ? 0 : (tok >= d.A ? d.A - 1 : tok);
    }
    toks[tid] = tok;
  }
  __syncthreads();
  // d mlp.0.weight[o][c*64 + k] = sum_b dh[b][o] in[b][k]   (rows >= B of both are zero)
  {
    const f32x16 acc = mfma64([&](int m, int k) { return dh[k][m]; }, [&](int k, int n) { return in[k][n]; }, 0.f);
    // Adam on this slice right here: mlp.0.weight is 96 % of the parameters, and its gradient would otherwise be
    // written out and read back by tr_adam (8 of the 28 bytes per parameter that pass moves).  The slice was copied to
    // LDS (ws) above, grad_in below reads that copy; no other workgroup touches this slice.
    const size_t base = (size_t)e * d.P + d.o_w0 + (size_t)c * H;
    quad_store(acc, [&](int o, int k, float gr) {
      const size_t i = base + (size_t)o * d.D + k;
      float pv = d.params[i], mv = d.m[i], vv = d.v[i];
      adam_update(pv, mv, vv, gr, lr_over_bc1, bc2_sqrt, beta1, beta2, eps);
      d.m[i] = mv; d.v[i] = vv; d.params[i] = pv;
    });
  }
  if (c < d.n_read) {
    __syncthreads();
    grad_in(in, dh, ws, in, false, B);   // in := d(embedding of the token at this position)
    // through the embedder: per-token sums in batch order (deterministic), bias = sum over the batch
    float* pe = d.part_e + ((size_t)e * d.n_read + c) * 32 * H;
    {
      // thread (k, q) keeps the sums of tokens 8q .. 8q+7 in registers and walks the batch in order: the same adds in
      // the same order as a per-token accumulator (adding 0.0f to the others changes no bit), without a serial
      // read-modify-write chain through LDS on 64 threads.  Slot 31 (no token: A <= 31) is the bias row.
      const int k = tid & 63, q = tid >> 6;
      float acc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] = 0.f;
      for (int b = 0; b < B; ++b) {
        const float v = in[b][k];
        const int tk = toks[b] - 8 * q;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += (tk == u || (q == 3 && u == 7)) ? v : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) pe[(8 * q + u) * H + k] = acc[u];
    }
  } else {
    // time chunk: d temb[b][j] = sum_o dh[b][o] W0[o][j]
    __syncthreads();
    grad_in(in, dh, ws, in, false, B);   // in := d temb  (grad_in reads dh/ws only, then overwrites `in`)
    // d t1.weight[j][k] = sum_b dtemb[b][j] f[b][k]: the Fourier features come back from where tr_fwd0 stored them
    // (through ws, free now, with zero rows past the batch: the buffer holds nothing defined there)
    {
      const float* f = d.fbuf + (size_t)e * TB * H;
      for (int idx = tid; idx < TB * H; idx += 256) ws[idx >> 6][idx & 63] = (idx >> 6) < B ? f[idx] : 0.f;
      __syncthreads();
      const f32x16 acc = mfma64([&](int m, int k) { return in[k][m]; }, [&](int k, int n) { return ws[k][n]; }, 0.f);
      float* dW = g + d.o_t1w;
      quad_store(acc, [&](int j, int k, float v) { dW[j * H + k] = v; });
      if (tid < H) {
        float sb = 0.f;
        for (int b = 0; b < B; ++b) sb += in[b][tid];
        g[d.o_t1b + tid] = sb;
      }
    }
  }
}

// gradient of embedder parameter q of member e: the sum of the per-position partials in a fixed shape -- lane `part` of
// the eight lanes that share the parameter adds positions part, part + 8, ... in order (all its loads in flight at
// once), then the eight partial sums are combined by an xor tree.  (One thread per parameter walking all 119
// positions was one memory latency per eight terms and set the duration of the whole pass.)
__device__ __forceinline__ float adam_grad_emb(const TrDev& d, int e, int row, int k, int part) {
  const float* pe = d.part_e + (size_t)e * d.n_read * 32 * H;
  float gr = 0.f;
  int c = part;
  for (; c + 56 < d.n_read; c += 64) {
    float v8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v8[u] = pe[((size_t)(c + 8 * u) * 32 + row) * H + k];
#pragma unroll
    for (int u = 0; u < 8; ++u) gr += v8[u];
  }
  for (; c < d.n_read; c += 8) gr += pe[((size_t)c * 32 + row) * H + k];
  gr += __shfl_xor(gr, 1);
  gr += __shfl_xor(gr, 2);
  gr += __shfl_xor(gr, 4);
  return gr;
}
// Adam for everything except mlp.0.weight (96 % of the parameters: tr_bwd0 updates its slices where their gradient
// is formed, so that gradient is never written out and read back).  The first `emb_blocks` blocks take the embedder,
// whose gradient is a 119-term sum over the positions, eight lanes per parameter; the others one parameter per thread
// over the index space with mlp.0.weight cut out.
__global__ __launch_bounds__(256) void tr_adam(TrDev d, int emb_blocks, float lr_over_bc1, float bc2_sqrt, float beta1, float beta2, float eps) {
  const int emb_lo = d.o_ew, emb_hi = d.o_b0 < d.o_w0 ? d.o_b0 : d.o_w0;
  if ((int)blockIdx.x < emb_blocks) {
    const int n_emb = emb_hi - emb_lo;
    const int j = ((int)blockIdx.x * 256 + threadIdx.x) >> 3, part = threadIdx.x & 7;   // eight lanes per parameter
    if (j >= d.E * n_emb) return;       // (whole groups of eight leave together: 256 and the shuffles stay aligned)
    // neighbouring lane groups take neighbouring hidden units k of one token row: their partials are neighbours in
    // part_e ([position][row][k]), where walking the parameters in blob order ([k][token]) touched a line per lane
    const int e = j / n_emb, j2 = j % n_emb;
    const int row = j2 < H * d.A ? j2 / H : 31, k = j2 % H;
    const int q = row < 31 ? d.o_ew + k * d.A + row : d.o_eb + k;
    const size_t i = (size_t)e * d.P + q;
    const float gr = adam_grad_emb(d, e, row, k, part);
    if (part == 0) adam_update(d.params[i], d.m[i], d.v[i], gr, lr_over_bc1, bc2_sqrt, beta1, beta2, eps);
    return;
  }
  const int w0_n = H * d.D, rest = d.P - w0_n;
  const int r = ((int)blockIdx.x - emb_blocks) * 256 + threadIdx.x;
  if (r >= d.E * rest) return;
  const int e = r / rest;
  int q = r % rest;
  if (q >= d.o_w0) q += w0_n;
  // (GaussianFourierProjection.W, q < o_t1w: requires_grad=False; the embedder: above)
  if (q < d.o_t1w || (q >= emb_lo && q < emb_hi)) return;
  const size_t i = (size_t)e * d.P + q;
  adam_update(d.params[i], d.m[i], d.v[i], d.grads[i], lr_over_bc1, bc2_sqrt, beta1, beta2, eps);
}

}  // namespace

struct sgpo_mlp_trainer {
  TrDev d;
  int* read_pos_dev;
  int max_read_pos;
  long long step;
  size_t mid_lds;
};

extern "C" size_t sgpo_mlp_trainer_param_count(int n_read, int alphabet, int time_conditioned) {
  const size_t h = H, A = alphabet, D = h * (n_read + (time_conditioned ? 1 : 0));
  return h / 2 + h * h + h + h * A + h + h * D + h + 3 * (h * h + h) + h + 1;
}

extern "C" void sgpo_mlp_trainer_destroy(sgpo_mlp_trainer* t) {
  if (!t) return;
  void* bufs[] = {t->d.params, t->d.grads, t->d.m, t->d.v, t->d.part_h, t->d.dh, t->d.fbuf, t->d.part_e, t->read_pos_dev};
  for (void* b : bufs) (void)hipFree(b);
  delete t;
}

extern "C" int sgpo_mlp_trainer_create(const float* blobs, int n_members, int n_read, const int* read_pos, int alphabet,
                                       int hidden, int time_conditioned, sgpo_mlp_trainer** out) {
  SGPO_REQUIRE(blobs && read_pos && out && n_members > 0 && n_read > 0, "mlp_trainer_create: bad arguments");
  SGPO_REQUIRE(hidden == H, "mlp_trainer_create: hidden_dim %d unsupported (64 only)", hidden);
  SGPO_REQUIRE(alphabet > 0 && alphabet <= 31, "mlp_trainer_create: alphabet %d (<= 31)", alphabet);
  sgpo_mlp_trainer* t = new (std::nothrow) sgpo_mlp_trainer();
  SGPO_REQUIRE(t, "mlp_trainer_create: out of host memory");
  TrDev& d = t->d;
  d = TrDev{};
  t->read_pos_dev = nullptr;
  d.E = n_members; d.n_read = n_read; d.A = alphabet; d.time_cond = time_conditioned ? 1 : 0;
  d.C = n_read + d.time_cond;
  d.D = H * d.C;
  d.P = (int)sgpo_mlp_trainer_param_count(n_read, alphabet, time_conditioned);
  int o = 0;
  d.o_tw = o; o += H / 2;
  d.o_t1w = o; o += H * H; d.o_t1b = o; o += H;
  d.o_ew = o; o += H * alphabet; d.o_eb = o; o += H;
  d.o_w0 = o; o += H * d.D; d.o_b0 = o; o += H;
  d.o_w1 = o; o += H * H; d.o_b1 = o; o += H;
  d.o_w2 = o; o += H * H; d.o_b2 = o; o += H;
  d.o_c0 = o; o += H * H; d.o_c0b = o; o += H;
  d.o_c2 = o; o += H; d.o_c2b = o; o += 1;
  const size_t EP = (size_t)d.E * d.P;
#define HIPF(e) do { hipError_t _e = (e); if (_e != hipSuccess) { sgpo_set_error("%s: %s", #e, hipGetErrorString(_e)); sgpo_mlp_trainer_destroy(t); return SGPO_ERR_HIP; } } while (0)
  HIPF(hipMalloc((void**)&d.params, EP * 4));
  HIPF(hipMalloc((void**)&d.grads, EP * 4));
  HIPF(hipMalloc((void**)&d.m, EP * 4));
  HIPF(hipMalloc((void**)&d.v, EP * 4));
  HIPF(hipMalloc((void**)&d.part_h, (size_t)d.E * d.C * TB * H * 4));
  HIPF(hipMalloc((void**)&d.dh, (size_t)d.E * TB * H * 4));
  HIPF(hipMalloc((void**)&d.fbuf, (size_t)d.E * TB * H * 4));
  HIPF(hipMalloc((void**)&d.part_e, (size_t)d.E * n_read * 32 * H * 4));
  HIPF(hipMalloc((void**)&t->read_pos_dev, (size_t)n_read * sizeof(int)));
  HIPF(hipMemcpy(t->read_pos_dev, read_pos, (size_t)n_read * sizeof(int), hipMemcpyHostToDevice));
  t->max_read_pos = 0;
  for (int i = 0; i < n_read; ++i) {
    if (read_pos[i] < 0) { sgpo_set_error("mlp_trainer_create: negative read position"); sgpo_mlp_trainer_destroy(t); return SGPO_ERR_ARG; }
    if (read_pos[i] > t->max_read_pos) t->max_read_pos = read_pos[i];
  }
  d.read_pos = t->read_pos_dev;
  HIPF(hipMemcpy(d.params, blobs, EP * 4, hipMemcpyDeviceToDevice));
  HIPF(hipMemset(d.grads, 0, EP * 4));
  HIPF(hipMemset(d.m, 0, EP * 4));
  HIPF(hipMemset(d.v, 0, EP * 4));
  t->step = 0;
  t->mid_lds = (size_t)(4 * TB + 3 * H + 2 * TB) * PAD * 4 + 2 * TB * 4;   // 150 KB
  HIPF(hipFuncSetAttribute(reinterpret_cast<const void*>(tr_mid), hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->mid_lds));
#undef HIPF
  *out = t;
  return SGPO_OK;
}

static int trainer_step(sgpo_mlp_trainer* t, const int64_t* x, const float* tt, const float* y, int B, int L, double lr,
                        double beta1, double beta2, double eps, float* loss_out, hipStream_t s) {
  const TrDev& d = t->d;
  const int n_groups = (d.C + FWD_CHUNKS - 1) / FWD_CHUNKS;
  hipLaunchKernelGGL(tr_fwd0, dim3(n_groups, d.E), dim3(256), 0, s, d, x, tt, B, L);
  hipLaunchKernelGGL(tr_reduce, dim3((d.E * TB * H + 255) / 256), dim3(256), 0, s, d, B, n_groups);
  hipLaunchKernelGGL(tr_mid, dim3(d.E), dim3(256), t->mid_lds, s, d, y, B, n_groups, loss_out);
  t->step += 1;
  // torch/optim/adam.py _single_tensor_adam: the scalars are formed in double, applied to fp32 tensors
  const double bc1 = 1.0 - pow(beta1, (double)t->step), bc2 = 1.0 - pow(beta2, (double)t->step);
  hipLaunchKernelGGL(tr_bwd0, dim3(d.C, d.E), dim3(256), 0, s, d, x, tt, B, L, (float)(lr / bc1), (float)sqrt(bc2),
                     (float)beta1, (float)beta2, (float)eps);
  const int main_blocks = (d.E * (d.P - H * d.D) + 255) / 256;
  const int n_emb = (d.o_b0 < d.o_w0 ? d.o_b0 : d.o_w0) - d.o_ew;
  const int emb_blocks = (d.E * n_emb * 8 + 255) / 256;   // eight lanes per embedder parameter
  hipLaunchKernelGGL(tr_adam, dim3((unsigned)(main_blocks + emb_blocks)), dim3(256), 0, s, d, emb_blocks,
                     (float)(lr / bc1), (float)sqrt(bc2), (float)beta1, (float)beta2, (float)eps);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

extern "C" int sgpo_mlp_trainer_steps(sgpo_mlp_trainer* t, const int64_t* x, const float* tt, const float* y, int n_steps, int B,
                                      int L, double lr, double beta1, double beta2, double eps, float* loss_out, void* stream) {
  SGPO_REQUIRE(t && x && y && loss_out, "mlp_trainer_step: null argument");
  SGPO_REQUIRE(n_steps > 0, "mlp_trainer_steps: n_steps=%d", n_steps);
  SGPO_REQUIRE(B > 0 && B <= TB, "mlp_trainer_step: batch %d outside 1..%d", B, TB);
  SGPO_REQUIRE(!t->d.time_cond || tt, "mlp_trainer_step: a time-conditioned model needs t");
  SGPO_REQUIRE(t->max_read_pos < L, "mlp_trainer_step: value function reads position %d of a length-%d row", t->max_read_pos, L);
  const size_t E = t->d.E;
  for (int j = 0; j < n_steps; ++j) {
    const int rc = trainer_step(t, x + (size_t)j * E * B * L, tt ? tt + (size_t)j * E * B : nullptr, y + (size_t)j * E * B, B, L, lr,
                                beta1, beta2, eps, loss_out + (size_t)j * E, (hipStream_t)stream);
    if (rc) return rc;
  }
  return SGPO_OK;
}

extern "C" int sgpo_mlp_trainer_step(sgpo_mlp_trainer* t, const int64_t* x, const float* tt, const float* y, int B, int L,
                                     double lr, double beta1, double beta2, double eps, float* loss_out, void* stream) {
  return sgpo_mlp_trainer_steps(t, x, tt, y, 1, B, L, lr, beta1, beta2, eps, loss_out, stream);
}

extern "C" int sgpo_mlp_trainer_read(const sgpo_mlp_trainer* t, float* blobs_out, void* stream) {
  SGPO_REQUIRE(t && blobs_out, "mlp_trainer_read: null argument");
  SGPO_CHECK_HIP(hipMemcpyAsync(blobs_out, t->d.params, (size_t)t->d.E * t->d.P * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return SGPO_OK;
}
