// Fitness-oracle ensemble: the step right after the sampling path.
// Reference: oracle/inference_oracle.py:10-72 (ensemble mean, Hamming penalty, max(., 0)) and
// oracle/train_oracle.py:39-48,61-74 (one-hot over 20 amino acids, Linear -> ReLU -> Linear).
//
// The one-hot input makes fc1 a gather-sum: pre[j] = b1[j] + sum_i W1[j][20 i + tok_i].  The weights
// are stored transposed once, T[m][i][a][j], so a gather reads `hidden` contiguous floats.  Per
// sequence and member that is n_res * hidden * 4 bytes from the (MALL/L2-resident) table; the
// bound is cache bandwidth, not HBM (table 38 MB for CreiLOV x 10 members, input 8 B/token).
#include "../../include/sgpo_hip.h"
#include "common.h"
#include "prof.h"

struct sgpo_fitness {
  int n_members = 0, n_res = 0, hidden = 0;
  float* T = nullptr;    // [m][i][20][hidden]
  float* b1 = nullptr;   // [m][hidden]
  float* w2 = nullptr;   // [m][hidden]
  float* b2 = nullptr;   // [m]
};

#define FIT_AA 20

__global__ void fitness_pack_kernel(const float* __restrict__ blob, int n_members, int n_res, int hidden,
                                    float* __restrict__ T, float* __restrict__ b1, float* __restrict__ w2,
                                    float* __restrict__ b2) {
  const size_t in_dim = (size_t)FIT_AA * n_res;
  const size_t per = (size_t)hidden * in_dim + hidden + hidden + 1;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n_t = (size_t)n_members * in_dim * hidden;
  if (idx < n_t) {
    const int j = idx % hidden;
    const size_t col = (idx / hidden) % in_dim;
    const size_t m = idx / ((size_t)hidden * in_dim);
    T[idx] = blob[m * per + (size_t)j * in_dim + col];   // fc1.weight [hidden][in_dim]
  }
  if (idx < (size_t)n_members * hidden) {
    const size_t m = idx / hidden, j = idx % hidden;
    b1[idx] = blob[m * per + (size_t)hidden * in_dim + j];
    w2[idx] = blob[m * per + (size_t)hidden * in_dim + hidden + j];
  }
  if (idx < (size_t)n_members) b2[idx] = blob[idx * per + (size_t)hidden * in_dim + 2 * hidden];
}

extern "C" void sgpo_fitness_destroy(sgpo_fitness* m) {
  if (!m) return;
  (void)hipFree(m->T); (void)hipFree(m->b1); (void)hipFree(m->w2); (void)hipFree(m->b2);
  delete m;
}

extern "C" int sgpo_fitness_create(const float* blob, size_t n_floats, int n_members, int n_res, int hidden,
                                   sgpo_fitness** out) {
  SGPO_REQUIRE(blob && out, "fitness_create: null argument");
  SGPO_REQUIRE(n_members > 0 && n_res > 0 && hidden > 0 && hidden <= 512, "fitness_create: members=%d n_res=%d hidden=%d (hidden <= 512)",
               n_members, n_res, hidden);
  const size_t in_dim = (size_t)FIT_AA * n_res;
  const size_t per = (size_t)hidden * in_dim + 2 * (size_t)hidden + 1;
  SGPO_REQUIRE(n_floats == per * n_members, "fitness_create: blob has %zu floats, expected %zu", n_floats, per * n_members);
  sgpo_fitness* m = new sgpo_fitness();
  m->n_members = n_members; m->n_res = n_res; m->hidden = hidden;
#define HIPF(e) do { hipError_t _e = (e); if (_e != hipSuccess) { sgpo_set_error("%s: %s", #e, hipGetErrorString(_e)); sgpo_fitness_destroy(m); return SGPO_ERR_HIP; } } while (0)
  const size_t n_t = (size_t)n_members * in_dim * hidden;
  HIPF(hipMalloc((void**)&m->T, n_t * 4));
  HIPF(hipMalloc((void**)&m->b1, (size_t)n_members * hidden * 4));
  HIPF(hipMalloc((void**)&m->w2, (size_t)n_members * hidden * 4));
  HIPF(hipMalloc((void**)&m->b2, (size_t)n_members * 4));
  hipLaunchKernelGGL(fitness_pack_kernel, dim3((unsigned)((n_t + 255) / 256)), dim3(256), 0, 0, blob, n_members, n_res, hidden,
                     m->T, m->b1, m->w2, m->b2);
  HIPF(hipGetLastError());
  HIPF(hipDeviceSynchronize());
#undef HIPF
  *out = m;
  return SGPO_OK;
}

// one workgroup (128 threads) per (sequence, member): pred = fc2(relu(fc1(onehot)))
__global__ __launch_bounds__(128) void fitness_member_kernel(const float* __restrict__ T, const float* __restrict__ b1,
                                                            const float* __restrict__ w2, const float* __restrict__ b2,
                                                            const int64_t* __restrict__ tokens, int n_res, int hidden,
                                                            int N, float* __restrict__ pred /*[m][N]*/, int* __restrict__ bad) {
  extern __shared__ int tok[];
  __shared__ float part[2];
  const int n = blockIdx.x, m = blockIdx.y, tid = threadIdx.x;
  for (int i = tid; i < n_res; i += 128) {
    const int64_t t = tokens[(size_t)n * n_res + i];
    if ((t < 0 || t >= FIT_AA) && bad) *bad = 1;      // the reference's alphabet.index() raises ValueError here
    tok[i] = t < 0 ? 0 : (t >= FIT_AA ? FIT_AA - 1 : (int)t);
  }
  __syncthreads();
  float pre[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { const int j = tid + 128 * u; pre[u] = j < hidden ? b1[(size_t)m * hidden + j] : 0.f; }
  const float* Tm = T + (size_t)m * n_res * FIT_AA * hidden;
  for (int i = 0; i < n_res; ++i) {
    const float* row = Tm + ((size_t)i * FIT_AA + tok[i]) * hidden;
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int j = tid + 128 * u; if (j < hidden) pre[u] += row[j]; }
  }
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) { const int j = tid + 128 * u; if (j < hidden) acc = fmaf(fmaxf(pre[u], 0.f), w2[(size_t)m * hidden + j], acc); }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) part[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) pred[(size_t)m * N + n] = part[0] + part[1] + b2[m];
}

// y[n] = mean_m max(pred[m][n] * penalty(hamming(parent, x_n)), 0) in fp64 (numpy float64 in the reference)
__global__ void fitness_finalize_kernel(const float* __restrict__ pred, const int64_t* __restrict__ tokens,
                                        const int64_t* __restrict__ parent, int n_res, int n_members, int N, int cutoff,
                                        double rate, double* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double pen = 1.0;
  if (parent) {
    int hd = 0;
    for (int i = 0; i < n_res; ++i) hd += tokens[(size_t)n * n_res + i] != parent[i];
    if (hd > cutoff) pen = pow(rate, (double)(hd - cutoff));
  }
  double s = 0.0;
  for (int m = 0; m < n_members; ++m) s += fmax((double)pred[(size_t)m * N + n] * pen, 0.0);
  out[n] = s / n_members;
}

extern "C" size_t sgpo_fitness_workspace_bytes(const sgpo_fitness* m, int N) {
  if (!m || N <= 0) return 0;
  return (((size_t)m->n_members * N * sizeof(float)) + 255) & ~(size_t)255;
}

extern "C" int sgpo_fitness_forward(sgpo_fitness* m, const int64_t* tokens, int N, const int64_t* parent_tokens, int cutoff,
                                    double rate, double* out, int* bad_token_flag, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  SGPO_REQUIRE(m && tokens && out && workspace, "fitness_forward: null argument");
  SGPO_REQUIRE(N > 0, "fitness_forward: N=%d", N);
  if (workspace_bytes < sgpo_fitness_workspace_bytes(m, N)) {
    sgpo_set_error("fitness_forward: workspace %zu < required %zu bytes", workspace_bytes, sgpo_fitness_workspace_bytes(m, N));
    return SGPO_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* pred = (float*)workspace;
  {
    ProfScope prof_scope(PK_FITNESS, st, (double)N * m->n_members * m->n_res * m->hidden * 4.0);   // table bytes gathered
    hipLaunchKernelGGL(fitness_member_kernel, dim3(N, m->n_members), dim3(128), m->n_res * sizeof(int), st, m->T, m->b1, m->w2,
                       m->b2, tokens, m->n_res, m->hidden, N, pred, bad_token_flag);
  }
  hipLaunchKernelGGL(fitness_finalize_kernel, dim3((N + 255) / 256), dim3(256), 0, st, pred, tokens, parent_tokens, m->n_res,
                     m->n_members, N, cutoff, rate, out);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}
