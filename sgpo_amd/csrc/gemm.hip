// bf16 MFMA GEMM with fused epilogues -- see gemm.h for the design notes.
#include "gemm.h"
#include "prof.h"
#include <stdlib.h>

#define BN 128
#define BK 64
#define B_TILE_BYTES (BN * BK * 2)  // 16 KiB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// MTW = 32-row MFMA tiles per wave along M: tile = (64 * MTW) x 128.  MTW = 1 fills the GPU at small M
// (a few sequences per call); same K order and epilogue arithmetic, so the bits do not depend on the choice.
template <int EPI, int MTW>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p, GemmSeg sg, const bf16_t* zero16) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x (A tile + B tile)
  constexpr int BM = 64 * MTW, A_TILE_BYTES = BM * BK * 2, STAGE_BYTES = A_TILE_BYTES + B_TILE_BYTES;
  // LDS ring: 4 slots of 24 KiB (64-row tile) or 3 of 32 KiB; the wave counts its own LDS-DMA operations
  constexpr int NST = MTW == 1 ? 4 : 3, OPS = 2 * MTW + 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware block remap: blocks that share an XCD (id % 8) get a contiguous run of tiles,
  // N-tiles fastest, so an A row-panel is re-read from that XCD's L2.
  const int ntn = (p.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int m0 = (bid / ntn) * BM;
  const int n0 = (bid % ntn) * BN;

  // ---- per-lane staging coordinates (4 LDS-DMA instructions per operand per wave)
  const bf16_t* a_src[2 * MTW];
  int a_l[2 * MTW];
  const bf16_t* w_src[4];
#pragma unroll
  for (int j = 0; j < 2 * MTW; ++j) {
    const int i = wave * 2 * MTW + j;
    const int row = i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    int m = m0 + row;
    if (m > p.M - 1) m = p.M - 1;
    a_src[j] = p.A + (size_t)m * p.lda + c * 8;
    a_l[j] = m % p.L;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = wave * 4 + j;
    const int row = i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    int n = n0 + row;
    if (n > p.N - 1) n = p.N - 1;
    w_src[j] = p.W + (size_t)n * p.ldw + c * 8;
  }

  auto stage = [&](int buf, int kt) {
    const int kglob = kt * BK;
    const int seg = kglob / sg.seg_k;
    const int kin = kglob - seg * sg.seg_k;
    const int sh = sg.shift[seg];
    char* abase = smem + buf * STAGE_BYTES;
    char* bbase = abase + A_TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 2 * MTW; ++j) {
      const int i = wave * 2 * MTW + j;
      const bf16_t* src = a_src[j] + (ptrdiff_t)sh * p.lda + kin;
      const int lp = a_l[j] + sh;
      if (lp < 0 || lp >= p.L) src = zero16;  // zero padding at the sequence ends (conv taps)
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(abase + i * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = wave * 4 + j;
      __builtin_amdgcn_global_load_lds((gptr_t)(w_src[j] + kglob), (lptr_t)(bbase + i * 1024), 16, 0, 0);
    }
  };

  // products: v_mfma_f32_16x16x32_bf16 (the shape of the persistent kernel, gemm_p.hip: same K order -- two 32-wide slices per
  // stage -- and the same instruction, so the two kernels give the same bits).  A 32x32 block is four 16x16 products:
  // acc[a][b][2 mh + nh][q] = D[m local = a*32 + 16 mh + 4 l4 + q][n local = b*32 + 16 nh + l16]
  f32x4 acc[MTW][2][4];
#pragma unroll
  for (int a = 0; a < MTW; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes inside a tile): row*128 + ((c ^ ((row>>1)&7)) * 16); lane (l16, l4) reads the 8 k of chunk
  // 4 ks + l4 of row l16 of a 16-row group (rows 16 h + l16 of a 32-row block share the swizzle term)
  const int l16 = lane & 15, l4 = lane >> 4;
  const int sw = (l16 >> 1) & 7;
  const int a_off = (wr * 32 * MTW + l16) * 128, b_off = (wc * 64 + l16) * 128;

  const int nk = p.K / BK;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) stage(s, s);
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt has landed once at most `ahead` younger stages (OPS operations each) are outstanding
    const int ahead = min(NST - 2, nk - 1 - kt);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * OPS) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave's pieces of stage kt are visible; every read of stage kt-1 is done
    asm volatile("" ::: "memory");
    if (kt + NST - 1 < nk) {
      int ps = slot + NST - 1;
      if (ps >= NST) ps -= NST;
      stage(ps, kt + NST - 1);
    }
    const char* abase = smem + slot * STAGE_BYTES;
    const char* bbase = abase + A_TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ((ks * 4 + l4) ^ sw) << 4;
      bf16x8 af[MTW][2], bfr[2][2];
#pragma unroll
      for (int t = 0; t < MTW; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) af[t][h] = *reinterpret_cast<const bf16x8*>(abase + a_off + (t * 32 + h * 16) * 128 + c);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) bfr[t][h] = *reinterpret_cast<const bf16x8*>(bbase + b_off + (t * 32 + h * 16) * 128 + c);
#pragma unroll
      for (int a = 0; a < MTW; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
              acc[a][b][2 * mh + nh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a][mh], bfr[b][nh], acc[a][b][2 * mh + nh], 0, 0, 0);
    }
    if (++slot == NST) slot = 0;
  }

  // ---------------------------------------------------------------- epilogue
  const int ncol0 = n0 + wc * 64;
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mh = r >> 3, nh = (r >> 2) & 1, q = r & 3;
      const int m = m0 + wr * 32 * MTW + mt * 32 + 16 * mh + 4 * l4 + q;
      if (m >= p.M) continue;
      float v0 = acc[mt][0][2 * mh + nh][q], v1 = acc[mt][1][2 * mh + nh][q];
      const int lc = 16 * nh + l16;   // column inside the 32-column block
      const int na = ncol0 + lc, nb = ncol0 + 32 + lc;
      if constexpr (EPI == EPI_QKV_ROPE) {
        bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
        if (ncol0 < p.rope_cols) {  // wave-uniform: a whole head (64 cols) of q or k
          const int pos = m % p.L;
          const float cs = p.rope_cos[pos * (p.head_dim / 2) + lc];
          const float sn = p.rope_sin[pos * (p.head_dim / 2) + lc];
          float o0, o1;
          epi_rope(v0, v1, cs, sn, o0, o1);
          v0 = o0; v1 = o1;
        }
        C[(size_t)m * p.ldc + na] = f2bf(v0);
        C[(size_t)m * p.ldc + nb] = f2bf(v1);
      } else if constexpr (EPI == EPI_BF16 || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_ERF) {
        bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
        if (p.bias) { v0 += p.bias[na]; v1 += p.bias[nb]; }
        if constexpr (EPI == EPI_BIAS_GELU) { v0 = epi_gelu_tanh(v0); v1 = epi_gelu_tanh(v1); }
        if constexpr (EPI == EPI_BIAS_GELU_ERF) { v0 = epi_gelu_erf(v0); v1 = epi_gelu_erf(v1); }
        C[(size_t)m * p.ldc + na] = f2bf(v0);
        C[(size_t)m * p.ldc + nb] = f2bf(v1);
      } else if constexpr (EPI == EPI_GATE_RES || EPI == EPI_RES) {
        float* C = reinterpret_cast<float*>(p.C);
        const float b0 = p.bias ? p.bias[na] : 0.f, b1 = p.bias ? p.bias[nb] : 0.f;
        const float x0 = p.xin[(size_t)m * p.ldc + na], x1 = p.xin[(size_t)m * p.ldc + nb];
        if constexpr (EPI == EPI_GATE_RES) {
          int b = m / p.L;
          if (b > p.B - 1) b = p.B - 1;
          const float* g = p.gate + (size_t)b * p.gate_stride;
          C[(size_t)m * p.ldc + na] = epi_gate_res(v0, b0, g[na], x0);
          C[(size_t)m * p.ldc + nb] = epi_gate_res(v1, b1, g[nb], x1);
        } else {
          C[(size_t)m * p.ldc + na] = x0 + (v0 + b0);
          C[(size_t)m * p.ldc + nb] = x1 + (v1 + b1);
        }
      } else {  // EPI_BIAS_F32 / EPI_ACC_F32
        float* C = reinterpret_cast<float*>(p.C);
        if (na < p.N) {
          if (p.bias) v0 += p.bias[na];
          if constexpr (EPI == EPI_ACC_F32) v0 += C[(size_t)m * p.ldc + na];
          C[(size_t)m * p.ldc + na] = v0;
        }
        if (nb < p.N) {
          if (p.bias) v1 += p.bias[nb];
          if constexpr (EPI == EPI_ACC_F32) v1 += C[(size_t)m * p.ldc + nb];
          C[(size_t)m * p.ldc + nb] = v1;
        }
      }
    }
  }
}

int gemm_cu_count() {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t pr;
    n_cu = (hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
  }
  return n_cu;
}

static const bf16_t* g_zero16[16] = {nullptr};

int gemm_get_zero16(const bf16_t** out) {
  int dev = 0;
  SGPO_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) { sgpo_set_error("device index %d out of range", dev); return SGPO_ERR_ARG; }
  if (!g_zero16[dev]) {
    void* z = nullptr;
    SGPO_CHECK_HIP(hipMalloc(&z, 16384));  // also serves as an all-zero bias vector (N <= 4096)
    SGPO_CHECK_HIP(hipMemset(z, 0, 16384));
    g_zero16[dev] = reinterpret_cast<const bf16_t*>(z);
  }
  *out = g_zero16[dev];
  return SGPO_OK;
}

extern "C" int sgpo_gemm_init_device() {  // called from sgpo_init: allocates the zero page once
  const bf16_t* z;
  return gemm_get_zero16(&z);
}

int launch_gemm_bf16_seg(const GemmParams& p, int epi, int n_seg, int seg_k, const int* shifts, hipStream_t stream) {
  SGPO_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  SGPO_REQUIRE(p.K % BK == 0, "gemm: K=%d must be a multiple of %d", p.K, BK);
  SGPO_REQUIRE(n_seg >= 1 && n_seg <= 8 && seg_k % BK == 0 && n_seg * seg_k == p.K, "gemm: bad K segmentation");
  SGPO_REQUIRE(p.L > 0, "gemm: L must be positive");
  SGPO_REQUIRE((p.lda % 8) == 0 && (p.ldw % 8) == 0, "gemm: lda/ldw must be multiples of 8 elements (16-B rows)");
  if (epi != EPI_BIAS_F32 && epi != EPI_ACC_F32)
    SGPO_REQUIRE(p.N % 64 == 0, "gemm: N=%d must be a multiple of 64 for this epilogue", p.N);
  GemmSeg sg;
  sg.n_seg = n_seg; sg.seg_k = seg_k;
  for (int i = 0; i < 8; ++i) sg.shift[i] = (shifts && i < n_seg) ? shifts[i] : 0;
  const bf16_t* z = nullptr;
  int rc = gemm_get_zero16(&z);
  if (rc) return rc;
  {
    static int use_v1 = -1;
    if (use_v1 < 0) { const char* e = getenv("SGPO_GEMM"); use_v1 = (e && e[0] == 'v' && e[1] == '1') ? 1 : 0; }
    if (!use_v1) {
      rc = launch_gemm_persistent(p, epi, sg, z, stream);
      if (rc != SGPO_ERR_UNSUPPORTED) return rc;
    }
  }
  const int ntn = (p.N + BN - 1) / BN;
  // 64-row tiles while 128-row tiles would leave CUs idle
  const bool small = ((p.M + 127) / 128) * ntn < gemm_cu_count();
  const int bm = small ? 64 : 128;
  const int ntm = (p.M + bm - 1) / bm;
  dim3 grid(ntm * ntn), block(256);
  const size_t lds = (small ? 4 : 3) * ((size_t)bm * BK * 2 + B_TILE_BYTES);
  ProfScope prof_scope(PK_GEMM_BF16 + epi, stream, 2.0 * p.M * p.N * p.K);
#define LAUNCH(E)                                                                                   \
  case E: {                                                                                         \
    static bool attr_set = false;                                                                   \
    if (!attr_set) {                                                                                \
      SGPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<E, 2>),     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 98304));       \
      SGPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<E, 1>),     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 98304));       \
      attr_set = true;                                                                              \
    }                                                                                               \
    if (small) hipLaunchKernelGGL((gemm_bf16_kernel<E, 1>), grid, block, lds, stream, p, sg, z);    \
    else hipLaunchKernelGGL((gemm_bf16_kernel<E, 2>), grid, block, lds, stream, p, sg, z);          \
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
      sgpo_set_error("gemm: unknown epilogue %d", epi);
      return SGPO_ERR_ARG;
  }
#undef LAUNCH
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

int launch_gemm_bf16(const GemmParams& p, int epi, hipStream_t stream) {
  return launch_gemm_bf16_seg(p, epi, 1, p.K, nullptr, stream);
}
