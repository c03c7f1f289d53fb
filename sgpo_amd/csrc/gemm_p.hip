// Persistent bf16 MFMA GEMM for the tall-skinny shapes of the denoisers (M = B*L in the 10^5..10^6
// range, K in 512..2560, N in 512..2048):  C[M,N] = A[M,K] * W[N,K]^T  + fused epilogue.
//
// Why a second kernel: with K = 512 a tile has only 8 K-steps, so the one-tile-ahead kernel of
// gemm.hip pays an exposed L2/HBM latency per step and a prologue/epilogue bubble per tile
// (profiles/r1: 340-505 TFLOP/s).  Here
//   * one 512-thread workgroup per CU walks its list of 256x128 output tiles (persistent);
//   * the (tile, k-step) space is FLATTENED into one stream of 64-deep K-steps that flows through a
//     3-slot LDS ring (48 KiB per slot) filled by LDS-DMA two steps ahead, across tile boundaries,
//     behind a counted `s_waitcnt vmcnt(6)` and raw `s_barrier`s (never vmcnt(0) inside a tile);
//   * a tile's epilogue runs at the top of the NEXT tile's first step, before that step's prefetch
//     is issued, so its stores are older than the newest prefetch and the counted wait stays exact;
//   * residual operands (fp32 x) are prefetched into registers one K-step before they are needed;
//   * the MFMA is issued "swapped" (A operand = weight rows, B operand = activation rows) so a lane
//     owns ONE output row and 4 consecutive columns per register quad: epilogues store 8 B (bf16)
//     or 16 B (fp32) per lane instead of 2-4 B, and RoPE pairs (d, d+32) stay lane-local;
//   * tiles are dealt so that the N-tiles of one 256-row A panel run on workgroups of one XCD
//     (`blockIdx % 8` labels the XCD group): the panel is fetched from HBM once per XCD L2.
// LDS image and swizzle are those of gemm.hip (lane-linear DMA image, 16-B chunk XOR (row>>1)&7 on
// the source address and on ds_read_b128: conflict-free).
#include "gemm.h"
#include "prof.h"

#define PBM 256
#define PBN 128
#define PBK 64
#define NST 3
#define A_BYTES (PBM * PBK * 2)
#define B_BYTES (PBN * PBK * 2)
#define STAGE_BYTES (A_BYTES + B_BYTES)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ float gelu_tanh_p(float x) {
  float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  float e = __expf(2.0f * u);
  return 0.5f * x * (2.0f - 2.0f / (e + 1.0f));
}
__device__ __forceinline__ float gelu_erf_p(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }

__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  uint2 r;
  r.x = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
  r.y = (uint32_t)f2bf(c) | ((uint32_t)f2bf(d) << 16);
  return r;
}

template <int EPI, bool CONV>
__global__ __launch_bounds__(512) void gemm_p_kernel(GemmParams p, GemmSeg sg, const bf16_t* zero16, int ntm, int ntn,
                                                     int grouped) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // NST x (A slot | B slot)
  constexpr bool RES = (EPI == EPI_GATE_RES || EPI == EPI_RES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int nk = p.K / PBK;

  // ---- this workgroup's tile list
  int t_first, t_stride, n_my, xcd = 0;
  if (grouped) {  // gridDim.x == 256: XCD group x owns panels x, x+8, ...; 32 workgroups per group
    xcd = blockIdx.x & 7;
    const int loc = blockIdx.x >> 3, gx = gridDim.x >> 3;
    const int panels = (ntm - xcd + 7) >> 3;
    const int tiles_x = panels * ntn;
    t_first = loc; t_stride = gx;
    n_my = tiles_x > loc ? (tiles_x - loc + gx - 1) / gx : 0;
  } else {
    t_first = blockIdx.x; t_stride = gridDim.x;
    const int tot = ntm * ntn;
    n_my = tot > t_first ? (tot - t_first + t_stride - 1) / t_stride : 0;
  }
  if (n_my == 0) return;
  auto tile_origin = [&](int i, int& m0, int& n0) {
    const int j = t_first + i * t_stride;
    if (grouped) { m0 = ((j / ntn) * 8 + xcd) * PBM; n0 = (j % ntn) * PBN; }
    else { m0 = (j / ntn) * PBM; n0 = (j % ntn) * PBN; }
  };

  // ---- LDS-DMA staging of flattened step s = (tile i, k-step kt)
  auto stage = [&](int s, int slot) {
    const int i = s / nk, kt = s - i * nk;
    int m0, n0;
    tile_origin(i, m0, n0);
    const int kglob = kt * PBK;
    int kin = kglob, sh = 0;
    if constexpr (CONV) {
      const int seg = kglob / sg.seg_k;
      kin = kglob - seg * sg.seg_k;
      sh = sg.shift[seg];
    }
    char* abase = smem + slot * STAGE_BYTES;
    char* bbase = abase + A_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ii = wave * 4 + j;
      const int row = ii * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      int m = m0 + row;
      if (m > p.M - 1) m = p.M - 1;
      const bf16_t* src = p.A + ((size_t)m + sh) * p.lda + kin + c * 8;
      if constexpr (CONV) {
        const int lp = m % p.L + sh;
        if (lp < 0 || lp >= p.L) src = zero16;
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(abase + ii * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ii = wave * 2 + j;
      const int row = ii * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      int n = n0 + row;
      if (n > p.N - 1) n = p.N - 1;
      __builtin_amdgcn_global_load_lds((gptr_t)(p.W + (size_t)n * p.ldw + kglob + c * 8), (lptr_t)(bbase + ii * 1024), 16, 0, 0);
    }
  };

  // fragment offsets inside a slot
  int x_off[2], w_off[2], x_sw[2], w_sw[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int rx = wm * 64 + t * 32 + lr;   // activation row (M)
    const int rw = wn * 64 + t * 32 + lr;   // weight row (N)
    x_off[t] = rx * 128; x_sw[t] = (rx >> 1) & 7;
    w_off[t] = A_BYTES + rw * 128; w_sw[t] = (rw >> 1) & 7;
  }

  f32x16 acc[2][2];  // [nt][mt]; D[row = n local][col = m local]
  float4 res[RES ? 16 : 1];

  // ---- epilogue of tile (m0, n0): lane owns rows m(mt) and column quads n = n0 + wn*64 + nt*32 + 8g + 4lh
  auto epilogue = [&](int m0, int n0) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int m = m0 + wm * 64 + mt * 32 + lr;
      const bool mok = m < p.M;
      const int nb0 = n0 + wn * 64;
      if constexpr (EPI == EPI_QKV_ROPE) {
        const bool rope = nb0 < p.rope_cols;  // wave-uniform: this wave's 64 columns are one head of q or k
        const int pos = (mok ? m : 0) % p.L;
        bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = 8 * g + 4 * lh;
          float a0[4], a1[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { a0[q] = acc[0][mt][4 * g + q]; a1[q] = acc[1][mt][4 * g + q]; }
          if (rope) {
            const float4 cs = *reinterpret_cast<const float4*>(p.rope_cos + pos * (p.head_dim / 2) + d);
            const float4 sn = *reinterpret_cast<const float4*>(p.rope_sin + pos * (p.head_dim / 2) + d);
            const float c4[4] = {cs.x, cs.y, cs.z, cs.w}, s4[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float o0, o1;
              epi_rope(a0[q], a1[q], c4[q], s4[q], o0, o1);
              a0[q] = o0; a1[q] = o1;
            }
          }
          if (mok) {
            *reinterpret_cast<uint2*>(C + (size_t)m * p.ldc + nb0 + d) = pack4(a0[0], a0[1], a0[2], a0[3]);
            *reinterpret_cast<uint2*>(C + (size_t)m * p.ldc + nb0 + 32 + d) = pack4(a1[0], a1[1], a1[2], a1[3]);
          }
        }
      } else if constexpr (EPI == EPI_BIAS_F32 || EPI == EPI_ACC_F32) {
        float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = nb0 + nt * 32 + 8 * g + 4 * lh;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (mok && n + q < p.N) {
                float o = acc[nt][mt][4 * g + q] + p.bias[n + q];
                if constexpr (EPI == EPI_ACC_F32) o += C[(size_t)m * p.ldc + n + q];
                C[(size_t)m * p.ldc + n + q] = o;
              }
          }
      } else {
        // all small-vector operands first (one wait), then arithmetic and stores
        float4 bb[8], gg[8];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = nb0 + nt * 32 + 8 * g + 4 * lh;
            bb[nt * 4 + g] = *reinterpret_cast<const float4*>(p.bias + n);
            if constexpr (EPI == EPI_GATE_RES) {
              int b = m / p.L;
              if (b > p.B - 1) b = p.B - 1;
              gg[nt * 4 + g] = *reinterpret_cast<const float4*>(p.gate + (size_t)b * p.gate_stride + n);
            }
          }
        if (mok) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int n = nb0 + nt * 32 + 8 * g + 4 * lh;
              const float4 b4 = bb[nt * 4 + g];
              float v[4] = {acc[nt][mt][4 * g + 0] + b4.x, acc[nt][mt][4 * g + 1] + b4.y,
                            acc[nt][mt][4 * g + 2] + b4.z, acc[nt][mt][4 * g + 3] + b4.w};
              if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = epi_gelu_tanh(v[q]);
              }
              if constexpr (EPI == EPI_BIAS_GELU_ERF) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = epi_gelu_erf(v[q]);
              }
              if constexpr (RES) {
                const float4 r = res[(mt * 2 + nt) * 4 + g];
                float4 o;
                if constexpr (EPI == EPI_GATE_RES) {
                  const float4 g4 = gg[nt * 4 + g];   // v already holds acc + bias: fma(acc + bias, gate, xin)
                  o = make_float4(fmaf(v[0], g4.x, r.x), fmaf(v[1], g4.y, r.y), fmaf(v[2], g4.z, r.z), fmaf(v[3], g4.w, r.w));
                } else {
                  o = make_float4(r.x + v[0], r.y + v[1], r.z + v[2], r.w + v[3]);
                }
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n) = o;
              } else {
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n) =
                    pack4(v[0], v[1], v[2], v[3]);
              }
            }
        }
      }
    }
  };

  auto load_residual = [&](int m0, int n0) {
    if constexpr (RES) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        int m = m0 + wm * 64 + mt * 32 + lr;
        if (m > p.M - 1) m = p.M - 1;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * 64 + nt * 32 + 8 * g + 4 * lh;
            res[(mt * 2 + nt) * 4 + g] = *reinterpret_cast<const float4*>(p.xin + (size_t)m * p.ldc + n);
          }
      }
    }
  };

  const int total = n_my * nk;
  stage(0, 0);
  if (total > 1) stage(1, 1);
  int slot = 0;       // ring slot of step s
  int kt = 0, ti = 0; // k-step inside the tile, tile index in my list
  for (int s = 0; s < total; ++s) {
    const bool first_of_tile = (kt == 0);
    // ---- wait until step s has landed: younger ops = this wave's 6 DMA of step s+1 (none at the very end)
    if (first_of_tile && ti > 0) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // also the residual prefetch of the tile just finished
    } else if (s + 1 < total) {
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (first_of_tile) {
      if (ti > 0) {
        int pm0, pn0;
        tile_origin(ti - 1, pm0, pn0);
        epilogue(pm0, pn0);   // stores are issued BEFORE the prefetch below: older than it in the vmcnt queue
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    }
    if (s + 2 < total) {
      int ps = slot + 2; if (ps >= NST) ps -= NST;
      stage(s + 2, ps);
    }
    if (RES && kt == nk - 1) {   // residual tile into registers while the last K-step multiplies
      int m0, n0;
      tile_origin(ti, m0, n0);
      load_residual(m0, n0);
    }
    const char* base = smem + slot * STAGE_BYTES;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int c = kk * 2 + lh;
      bf16x8 xf[2], wf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        xf[t] = *reinterpret_cast<const bf16x8*>(base + x_off[t] + ((c ^ x_sw[t]) << 4));
        wf[t] = *reinterpret_cast<const bf16x8*>(base + w_off[t] + ((c ^ w_sw[t]) << 4));
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (++slot == NST) slot = 0;
    if (++kt == nk) { kt = 0; ++ti; }
  }
  if (RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int pm0, pn0;
  tile_origin(n_my - 1, pm0, pn0);
  epilogue(pm0, pn0);
}

int launch_gemm_persistent(const GemmParams& p, int epi, const GemmSeg& sg, const bf16_t* zero16, hipStream_t stream) {
  // shapes the vectorised epilogues need; anything else falls back to the generic kernel
  if (p.K % PBK != 0 || sg.seg_k % PBK != 0) return SGPO_ERR_UNSUPPORTED;
  if (epi != EPI_BIAS_F32 && epi != EPI_ACC_F32 && (p.N % 64 != 0 || p.ldc % 4 != 0)) return SGPO_ERR_UNSUPPORTED;
  if (epi == EPI_QKV_ROPE && (p.head_dim != 64 || p.rope_cols % 64 != 0)) return SGPO_ERR_UNSUPPORTED;
  if ((epi == EPI_GATE_RES || epi == EPI_RES) && (p.xin == nullptr || (p.gate == nullptr && epi == EPI_GATE_RES))) return SGPO_ERR_UNSUPPORTED;
  if (p.M < 1024) return SGPO_ERR_UNSUPPORTED;  // tiny problems: the 128x128 kernel fills more CUs
  const int ntm = (p.M + PBM - 1) / PBM, ntn = (p.N + PBN - 1) / PBN;
  const int tiles = ntm * ntn;
  int dev = 0, n_cu = 256;
  (void)hipGetDevice(&dev);
  static int cached_cu = 0;
  if (!cached_cu) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) == hipSuccess) cached_cu = pr.multiProcessorCount; else cached_cu = 256; }
  n_cu = cached_cu;
  const int grouped = (tiles > n_cu && n_cu % 8 == 0 && ntm >= 8) ? 1 : 0;
  const int grid = tiles > n_cu ? n_cu : tiles;
  const size_t lds = NST * STAGE_BYTES;
  GemmParams q = p;
  if (!q.bias) q.bias = reinterpret_cast<const float*>(zero16);  // zero page (16 KiB): N <= 4096
  if (q.N > 4096 && !p.bias) return SGPO_ERR_UNSUPPORTED;
  const bool conv = sg.n_seg > 1;
  ProfScope prof_scope(PK_GEMM_BF16 + epi, stream, 2.0 * p.M * p.N * p.K);
#define LAUNCH(E)                                                                                      \
  case E: {                                                                                            \
    static bool attr_set = false;                                                                      \
    if (!attr_set) {                                                                                   \
      SGPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_p_kernel<E, false>),       \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
      SGPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_p_kernel<E, true>),        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
      attr_set = true;                                                                                 \
    }                                                                                                  \
    if (conv)                                                                                          \
      hipLaunchKernelGGL((gemm_p_kernel<E, true>), dim3(grid), dim3(512), lds, stream, q, sg, zero16, ntm, ntn, grouped); \
    else                                                                                               \
      hipLaunchKernelGGL((gemm_p_kernel<E, false>), dim3(grid), dim3(512), lds, stream, q, sg, zero16, ntm, ntn, grouped); \
  } break;
  switch (epi) {
    LAUNCH(EPI_BF16)
    LAUNCH(EPI_QKV_ROPE)
    LAUNCH(EPI_GATE_RES)
    LAUNCH(EPI_BIAS_GELU)
    LAUNCH(EPI_BIAS_F32)
    LAUNCH(EPI_BIAS_GELU_ERF)
    LAUNCH(EPI_RES)
    LAUNCH(EPI_ACC_F32)
    default:
      return SGPO_ERR_UNSUPPORTED;
  }
#undef LAUNCH
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}
