// Second persistent bf16 MFMA GEMM (round 3 experiment -> DESIGN.md section 6): TWO independent 256-thread workgroups
// per CU instead of one 512-thread workgroup, so that the HBM-bound epilogue of one workgroup's tile overlaps the
// MFMA-bound K loop of the other's.  gemm_p.hip cannot overlap the two phases (both SIMD partners belong to the same tile
// and finish it together; a second accumulator set does not fit the register file).  Here each workgroup owns a whole
// 256 x 128 tile: 4 waves in a 2(M) x 2(N) layout, wave tile 128 x 64 = the accumulator layout of gemm_p.hip's CfgBig, so
// the epilogue arithmetic and its line-wise staging carry over unchanged.  What it costs: 1.5x the operand bytes per flop
// of the 256 x 256 tile, and BK = 32 (64-byte rows in the LDS image) so that a 3-slot ring fits 80 KiB per workgroup.
//
//   C[M,N] = A[M,K] * W[N,K]^T (+ epilogue), A / W bf16 row-major with K contiguous, fp32 accumulate,
//   v_mfma_f32_32x32x16_bf16 in the k order of gemm.hip / gemm_p.hip.  (It gave the accumulator bits of those two kernels when it
//   was measured; they have since moved to the 16x16x32 shape, this experiment has not: it agrees with them to fp32
//   summation order only.  Not on any product path.)
//
// Stream per workgroup: (tile, k-step) flattened; at step s the wave issues the LDS-DMA of step s + 2 into the slot of
// step s - 1, reads its fragments of step s, waits for its own DMA pieces of step s + 1 (counted vmcnt) and for its
// reads, ONE s_barrier, then 16 MFMAs.  RAW: every wave's pieces of step s + 1 have landed before the barrier of step
// s; WAR: the slot refilled at step s + 1 (slot of step s) was last read before that same barrier.
#include "gemm.h"
#include "prof.h"
#include <stdlib.h>

typedef const __attribute__((address_space(1))) void* gptr_q;
typedef __attribute__((address_space(3))) void* lptr_q;
typedef uint32_t u32x2_q __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_q __attribute__((ext_vector_type(4)));
typedef float f32x4_q __attribute__((ext_vector_type(4)));

namespace {
constexpr int QBM = 256, QBN = 128, QBK = 32, QNST = 3, QD = QNST - 1;
constexpr int QMT = 4, QNT = 2;                           // wave tile 128 x 64
constexpr int QROWB = QBK * 2;                            // 64-byte rows
constexpr int QA_BYTES = QBM * QROWB, QB_BYTES = QBN * QROWB, QSTAGE = QA_BYTES + QB_BYTES;   // 16 + 8 KiB
constexpr int QA_PW = QA_BYTES / 4096, QB_PW = QB_BYTES / 4096, QOPS = QA_PW + QB_PW;         // pieces per wave: 4 + 2
constexpr int QRING = QNST * QSTAGE;                      // 72 KiB

template <int N>
__device__ __forceinline__ void q_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ uint32_t q_lds_addr(const void* p) { return (uint32_t)(size_t)(lptr_q)(p); }
__device__ __forceinline__ void q_lds_write8(uint32_t addr, u32x2_q v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void q_lds_read_rows(u32x4_q (&o)[4], uint32_t a0, uint32_t a1) {
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\t"
      "ds_read_b128 %0, %4\n\t"
      "ds_read_b128 %1, %4 offset:1024\n\t"
      "ds_read_b128 %2, %5\n\t"
      "ds_read_b128 %3, %5 offset:1024\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(a0), "v"(a1)
      : "memory");
}
__device__ __forceinline__ void q_lds_read4x16(f32x4_q (&o)[4], uint32_t addr) {
  asm volatile(
      "ds_read_b128 %0, %4\n\t"
      "ds_read_b128 %1, %4 offset:32\n\t"
      "ds_read_b128 %2, %4 offset:64\n\t"
      "ds_read_b128 %3, %4 offset:96\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(addr)
      : "memory");
}
__device__ __forceinline__ u32x2_q q_pack4(float a, float b, float c, float d) {
  u32x2_q r;
  r.x = pack_bf16x2(a, b);
  r.y = pack_bf16x2(c, d);
  return r;
}
}  // namespace

// EPI: EPI_BF16 (bias), EPI_BIAS_GELU.  mode bit 0: skip the epilogue (diagnostic: K loop alone).
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_q_kernel(GemmParams p, int ntm, int ntn, int grouped, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // QNST x (A slot | B slot) | bias table | staging
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int nk = p.K / QBK;
  float* auxf = reinterpret_cast<float*>(smem + QRING);          // bias [N]

  // ---- tile list: the N tiles of one A row panel on workgroups of one XCD (blockIdx & 7), as gemm_p.hip
  int t_first, t_stride, n_my, xcd = 0;
  if (grouped) {
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
    if (grouped) { m0 = ((j / ntn) * 8 + xcd) * QBM; n0 = (j % ntn) * QBN; }
    else { m0 = (j / ntn) * QBM; n0 = (j % ntn) * QBN; }
  };

  for (int i = tid; i < p.N; i += 256) auxf[i] = p.bias[i];
  __syncthreads();

  // ---- LDS-DMA staging: a piece = 16 rows x 64 B (1 KiB per wave instruction); lane -> row lane / 4, chunk lane % 4,
  // source chunk XORed with (row >> 2) & 3 (the same involution on the fragment reads: conflict-free ds_read_b128)
  int st_i = 0, st_kt = 0, st_m0, st_n0;
  tile_origin(0, st_m0, st_n0);
  // per-lane byte offsets of the pieces inside a full tile: constant for the launch
  uint32_t a_off[QA_PW], w_off[QB_PW];
#pragma unroll
  for (int j = 0; j < QA_PW; ++j) {
    const int row = (wave * QA_PW + j) * 16 + (lane >> 2);
    a_off[j] = ((uint32_t)row * (uint32_t)p.lda + (((lane & 3) ^ ((row >> 2) & 3)) * 8)) * 2;
  }
#pragma unroll
  for (int j = 0; j < QB_PW; ++j) {
    const int row = (wave * QB_PW + j) * 16 + (lane >> 2);
    w_off[j] = ((uint32_t)row * (uint32_t)p.ldw + (((lane & 3) ^ ((row >> 2) & 3)) * 8)) * 2;
  }
  auto stage = [&](int slot) {
    const int kglob = st_kt * QBK;
    char* abase = smem + slot * QSTAGE;
    char* bbase = abase + QA_BYTES;
    if (st_m0 + QBM <= p.M && st_n0 + QBN <= p.N) {
      const char* ag = reinterpret_cast<const char*>(p.A + (size_t)st_m0 * p.lda + kglob);
      const char* wg = reinterpret_cast<const char*>(p.W + (size_t)st_n0 * p.ldw + kglob);
#pragma unroll
      for (int j = 0; j < QA_PW; ++j)
        __builtin_amdgcn_global_load_lds((gptr_q)(ag + a_off[j]), (lptr_q)(abase + (wave * QA_PW + j) * 1024), 16, 0, 0);
#pragma unroll
      for (int j = 0; j < QB_PW; ++j)
        __builtin_amdgcn_global_load_lds((gptr_q)(wg + w_off[j]), (lptr_q)(bbase + (wave * QB_PW + j) * 1024), 16, 0, 0);
      if (++st_kt == nk) {
        st_kt = 0;
        if (++st_i < n_my) tile_origin(st_i, st_m0, st_n0);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < QA_PW; ++j) {
      const int ii = wave * QA_PW + j;
      const int row = ii * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      int m = st_m0 + row;
      if (m > p.M - 1) m = p.M - 1;
      __builtin_amdgcn_global_load_lds((gptr_q)(p.A + (size_t)m * p.lda + kglob + c * 8), (lptr_q)(abase + ii * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < QB_PW; ++j) {
      const int ii = wave * QB_PW + j;
      const int row = ii * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      int n = st_n0 + row;
      if (n > p.N - 1) n = p.N - 1;
      __builtin_amdgcn_global_load_lds((gptr_q)(p.W + (size_t)n * p.ldw + kglob + c * 8), (lptr_q)(bbase + ii * 1024), 16, 0, 0);
    }
    if (++st_kt == nk) {
      st_kt = 0;
      if (++st_i < n_my) tile_origin(st_i, st_m0, st_n0);
    }
  };

  const int sw = (lr >> 2) & 3;
  const int a_lane = (wm * QMT * 32 + lr) * QROWB;
  const int b_lane = QA_BYTES + (wn * QNT * 32 + lr) * QROWB;
  int cx[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) cx[kk] = ((kk * 2 + lh) ^ sw) << 4;

  f32x16 acc[QNT][QMT];

  // ---- epilogue of tile (m0, n0): a lane owns row lr of each 32-row block and the column quads nb0 + nt*32 + 8g + 4lh;
  // a 32 x 64 bf16 block (4 KiB) is staged through the wave's own LDS area so that every global store instruction writes
  // whole 128-byte lines (image as gemm_p.hip store_block: row r at (r & 15) * 128 in half r >> 4, 16-B chunk ch at
  // ch ^ (r & 7), the 8-B halves of a chunk swapped when r & 8)
  auto epilogue = [&](int m0, int n0, int fslot) {
    const int nb0 = n0 + wn * QNT * 32;
    // staging = the wave's own A pieces (4 KiB) of the FREE ring slot: the slot of the tile's last step, which this wave
    // itself refills at the next step -- no other wave touches these bytes, no barrier is needed
    const uint32_t h0 = q_lds_addr(smem + fslot * QSTAGE + wave * (QA_PW * 1024)), h1 = h0 + 2048;
    f32x4_q bb[QNT][4];
#pragma unroll
    for (int nt = 0; nt < QNT; ++nt) q_lds_read4x16(bb[nt], q_lds_addr(auxf + nb0 + nt * 32 + 4 * lh));
#pragma unroll
    for (int mt = 0; mt < QMT; ++mt) {
      const int m_blk = m0 + wm * QMT * 32 + mt * 32;
#pragma unroll
      for (int nt = 0; nt < QNT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4];
          const float bq[4] = {bb[nt][g].x, bb[nt][g].y, bb[nt][g].z, bb[nt][g].w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float a = acc[nt][mt][4 * g + q] + bq[q];
            v[q] = (EPI == EPI_BIAS_GELU) ? epi_gelu_tanh(a) : a;
          }
          const uint32_t wbase = ((lr >> 4) ? h1 : h0) + (lr & 15) * 128 + ((lh ^ ((lr >> 3) & 1)) << 3);
          q_lds_write8(wbase + (((nt * 4 + g) ^ (lr & 7)) << 4), q_pack4(v[0], v[1], v[2], v[3]));
        }
      const int r8 = lane >> 3, c8 = lane & 7;
      const uint32_t roff = r8 * 128 + ((c8 ^ r8) << 4);
      u32x4_q o[4];
      q_lds_read_rows(o, h0 + roff, h1 + roff);
      bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + (size_t)(m_blk + r8) * p.ldc + nb0 + c8 * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x4_q v = o[i];
        if (i & 1) v = u32x4_q{o[i].z, o[i].w, o[i].x, o[i].y};
        if (m_blk + i * 8 + r8 < p.M && nb0 + c8 * 8 < p.N)
          __builtin_nontemporal_store(v, reinterpret_cast<u32x4_q*>(cp + (size_t)(i * 8) * p.ldc));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- main stream
  const int total = n_my * nk;
#pragma unroll
  for (int d = 0; d < QD; ++d)
    if (d < total) stage(d);
  if (total > 1) q_wait_vm<QOPS>(); else q_wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  int slot = 0, s = 0;
  bool st_pending = false;
  for (int ti = 0; ti < n_my; ++ti) {
#pragma unroll
    for (int a = 0; a < QNT; ++a)
#pragma unroll
      for (int b = 0; b < QMT; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int kt = 0; kt < nk; ++kt) {
      if (s + QD < total) {
        int ps = slot + QD; if (ps >= QNST) ps -= QNST;
        stage(ps);
      }
      const char* base = smem + slot * QSTAGE;
      bf16x8 xf[2][QMT], wf[2][QNT];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int t = 0; t < QNT; ++t) wf[kk][t] = *reinterpret_cast<const bf16x8*>(base + b_lane + cx[kk] + t * 32 * QROWB);
#pragma unroll
        for (int t = 0; t < QMT; ++t) xf[kk][t] = *reinterpret_cast<const bf16x8*>(base + a_lane + cx[kk] + t * 32 * QROWB);
      }
      // own pieces of step s + 1 landed: younger in the queue are the pieces of step s + 2 issued above (if any) and, at
      // the first step after a full tile's epilogue, that epilogue's 16 store instructions (issued before them)
      if (s + 1 < total) {
        if (st_pending && kt == 0) { if (s + QD < total) q_wait_vm<QOPS + 16>(); else q_wait_vm<16>(); }
        else if (s + QD < total) q_wait_vm<QOPS>();
        else q_wait_vm<0>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int nt = 0; nt < QNT; ++nt)
#pragma unroll
          for (int mt = 0; mt < QMT; ++mt)
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk][nt], xf[kk][mt], acc[nt][mt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (++slot == QNST) slot = 0;
      ++s;
    }
    int m0, n0;
    tile_origin(ti, m0, n0);
    if (!(mode & 1)) {
      int fslot = slot + QNST - 1; if (fslot >= QNST) fslot -= QNST;
      epilogue(m0, n0, fslot);
      // a full tile issues exactly 16 store instructions per wave and no load: the next step's wait counts them; an edge
      // tile (rows or columns predicated off, an unknown number of stores) is drained
      st_pending = (m0 + QBM <= p.M && n0 + QBN <= p.N);
      if (!st_pending) q_wait_vm<0>();
    } else {
      float sink = 0.f;
#pragma unroll
      for (int a = 0; a < QNT; ++a)
#pragma unroll
        for (int b = 0; b < QMT; ++b) sink += acc[a][b][0] + acc[a][b][15];
      if (sink == 123.456f) reinterpret_cast<float*>(p.C)[0] = sink;   // keeps the accumulators alive
    }
  }
}

// C bf16 [M][ldc] = epi(A W^T + bias).  epi: EPI_BF16 or EPI_BIAS_GELU.  Returns SGPO_ERR_UNSUPPORTED for other shapes.
int launch_gemm_q(const GemmParams& p, int epi, hipStream_t stream) {
  if (epi != EPI_BF16 && epi != EPI_BIAS_GELU) return SGPO_ERR_UNSUPPORTED;
  if (p.K % QBK != 0 || p.N % 64 != 0 || p.lda % 8 != 0 || p.ldw % 8 != 0 || p.ldc % 8 != 0 || p.N > 4096 || p.M < 1) return SGPO_ERR_UNSUPPORTED;
  const int n_cu = gemm_cu_count();
  const int ntm = (p.M + QBM - 1) / QBM, ntn = (p.N + QBN - 1) / QBN;
  const int tiles = ntm * ntn;
  const int wgs = 2 * n_cu;
  const int grouped = (tiles > wgs && wgs % 8 == 0 && ntm >= 8) ? 1 : 0;
  const int grid = tiles > wgs ? wgs : tiles;
  const size_t lds = QRING + ((p.N * 4 + 15) & ~15);     // 72 KiB ring + bias: two workgroups share the CU's 160 KiB
  if (lds > 81920) return SGPO_ERR_UNSUPPORTED;
  GemmParams q = p;
  if (!q.bias) {
    const bf16_t* z = nullptr;
    if (int rc = gemm_get_zero16(&z)) return rc;
    q.bias = reinterpret_cast<const float*>(z);
  }
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("SGPO_GEMMQ_MODE"); mode = e ? atoi(e) : 0; }
  static bool attr_set = false;
  if (!attr_set) {
    SGPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_q_kernel<EPI_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
    SGPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_q_kernel<EPI_BIAS_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
    attr_set = true;
  }
  ProfScope prof_scope(PK_GEMM_BF16 + epi, stream, 2.0 * p.M * p.N * p.K);
  if (epi == EPI_BF16) hipLaunchKernelGGL(gemm_q_kernel<EPI_BF16>, dim3(grid), dim3(256), lds, stream, q, ntm, ntn, grouped, mode);
  else hipLaunchKernelGGL(gemm_q_kernel<EPI_BIAS_GELU>, dim3(grid), dim3(256), lds, stream, q, ntm, ntn, grouped, mode);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}
