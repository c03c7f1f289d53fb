// Persistent bf16 MFMA GEMM for the tall-skinny shapes of the denoisers (M = B*L in the 10^5..10^6
// range, K in 512..2560, N in 512..2048):  C[M,N] = A[M,K] * W[N,K]^T  + fused epilogue.
//
// Why a second kernel: with K = 512 a tile has only 8 K-steps, so the one-tile-ahead kernel of
// gemm.hip pays an exposed L2/HBM latency per step and a prologue/epilogue bubble per tile
// (profiles/r1: 340-505 TFLOP/s).  Here
//   * one 512-thread workgroup per CU walks its list of output tiles (persistent);
//   * the (tile, k-step) space is FLATTENED into one stream of K-steps that flows through an LDS
//     ring filled by LDS-DMA D = NST-1 steps ahead, across tile boundaries, behind a counted
//     `s_waitcnt vmcnt(N)` and raw `s_barrier`s (never vmcnt(0) inside the stream);
//   * a tile's epilogue runs at the top of the NEXT tile's first step, before that step's prefetch
//     is issued.  Its stores sit in the same in-order VM queue as the prefetches, so for the D-1
//     steps that follow the wait count includes them (exact count: the epilogue of a full tile
//     issues a compile-time number of store instructions and no loads -- bias and the RoPE table
//     are copied to LDS once per launch);
//   * the MFMA (v_mfma_f32_16x16x32_bf16 since round 3: the chip holds a higher clock on it than on
//     32x32x16 at equal cycles per flop) is issued "swapped" (A operand = weight rows, B operand =
//     activation rows) so a register quad is 4 consecutive columns of ONE output row: a lane owns two
//     rows and two column quads of every 32x32 block, epilogues move 8 B (bf16) or 16 B (fp32) per
//     lane and quad instead of 2-4 B, and RoPE pairs (d, d+32) stay lane-local;
//   * tiles are dealt so that the N-tiles of one A row panel run on workgroups of one XCD
//     (`blockIdx % 8` labels the XCD group): the panel is fetched from HBM once per XCD L2.
// Two tile configurations (the staging traffic per flop, not the MFMA rate, bounds these shapes:
// PMC shows ~11 B/clk/CU of LDS-DMA at 24 % MFMA utilisation, so the bigger tile wins wherever the
// epilogue has no fp32 residual tile to carry in registers):
//   CfgBig 256x256, BK 64, 2-slot ring (128 KiB), waves 2(M) x 4(N), wave tile 128x64 (BK 32 with a
//          4-slot ring measured 5 % slower: its 64-B rows make twice the L2 requests per DMA piece,
//          ~175 instead of ~105 cycles of issue time per piece)
//   CfgRes 256x128, BK 64, 3-slot ring (144 KiB), waves 4(M) x 2(N), wave tile 64x64; the fp32
//          residual tile is prefetched into registers one K-step before it is needed.
// LDS image and swizzle follow gemm.hip (lane-linear DMA image; the 16-B chunk index is XORed with
// (row>>1)&7 for 128-B rows or (row>>2)&3 for 64-B rows on the source address and on ds_read_b128:
// conflict-free for the four 16-lane groups of ds_read_b128).
#include "gemm.h"
#include "ln_stats.h"
#include "prof.h"
#include <stdlib.h>
#include <type_traits>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

struct CfgBig { static constexpr int WM = 2, WN = 4, MT = 4, NT = 2, BK = 64, NST = 2, PH = 2; };
struct CfgRes { static constexpr int WM = 4, WN = 2, MT = 2, NT = 2, BK = 64, NST = 3, PH = 2; };
// Full-row tile for N = 512: 128 rows x 512 columns, BK 64, 2-slot ring (the whole 160 KiB), waves 1(M) x 8(N), wave tile
// 128x64 as in CfgBig.  One workgroup owns whole output rows, so an epilogue can normalise over the row (EPI_LN_GELU): the
// eight waves exchange per-row partial statistics through the A region of the free ring slot between two extra barriers.
// Operand traffic per flop is 1.25x that of 256x256; the K loop alone measured 647 vs 608 us on the out-proj/fc2 shapes
// (DESIGN.md section 6).  No LDS is left for the bias table: bias and the LayerNorm affine come from global memory (L2).
struct CfgRow { static constexpr int WM = 1, WN = 8, MT = 4, NT = 2, BK = 64, NST = 2, PH = 2; };
#ifdef SGPO_GEMM_ABLATE
// Diagnostic build only (SGPO_GEMM_TILE=h): 64 x 512 tile, K loop only.  Its accumulators are 64 registers per lane, so two
// tiles could be in flight per wave (the epilogue of one sliced into the K steps of the next); this prices the K loop
// of such a design: 2.25x the operand bytes per flop of 256 x 256.
struct CfgHalf { static constexpr int WM = 1, WN = 8, MT = 2, NT = 2, BK = 64, NST = 2, PH = 2; };
template <typename C> struct is_row_cfg : std::integral_constant<bool, std::is_same<C, CfgRow>::value || std::is_same<C, CfgHalf>::value> {};
#else
template <typename C> struct is_row_cfg : std::is_same<C, CfgRow> {};
#endif

enum { PF_EXACT = 1 };
#ifndef SGPO_G1_PRIO
#define SGPO_G1_PRIO 1
#endif
#define LDS_MAX_BYTES 163840

#ifdef SGPO_GEMM_ABLATE   // diagnostic build only (tools/dbg): skip parts of the kernel at run time
#define ABL(bit) ((dbg & (bit)) != 0)
#else
#define ABL(bit) false
#endif

#ifdef SGPO_GEMM_STAMP   // diagnostic build only (tools/dbg/gemm_stamps.py): per-wave cycle totals by loop section
__device__ unsigned long long g_stamps[1024 * 8 * 8];
#define STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t = __builtin_amdgcn_s_memtime();
#define STAMP(i) do { const unsigned long long st_n = __builtin_amdgcn_s_memtime(); st_acc[i] += st_n - st_t; st_t = st_n; } while (0)
#define STAMP_TAKE(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define STAMP_PUT(i, v) do { st_acc[i] += (v) - st_t; st_t = (v); } while (0)
#define STAMP_FLUSH do { if ((dbg & 8) && lane == 0) for (int i = 0; i < 8; ++i) g_stamps[(blockIdx.x * 8 + wave) * 8 + i] = st_acc[i]; } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_TAKE(v)
#define STAMP_PUT(i, v)
#define STAMP_FLUSH
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ u32x2_t pack4(float a, float b, float c, float d) {
  u32x2_t r;
  r.x = pack_bf16x2(a, b);
  r.y = pack_bf16x2(c, d);
  return r;
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// 4 x 16 B from LDS at addr + 32 i, read and waited for inside one asm statement.  The compiler does not
// see these as LDS loads, so it does not drain the LDS-DMA queue (vmcnt(0)) in front of them, which it
// does for any ordinary ds_read it cannot prove disjoint from the DMA image.
__device__ __forceinline__ void lds_read4x16(f32x4_t (&o)[4], uint32_t addr) {
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
// same with four separate addresses (swizzled table rows)
__device__ __forceinline__ void lds_read4(f32x4_t (&o)[4], uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3) {
  asm volatile(
      "ds_read_b128 %0, %4\n\t"
      "ds_read_b128 %1, %5\n\t"
      "ds_read_b128 %2, %6\n\t"
      "ds_read_b128 %3, %7\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
      : "memory");
}
// 2 x 16 B at two addresses / at addr and addr + 64
__device__ __forceinline__ void lds_read2(f32x4_t (&o)[2], uint32_t a0, uint32_t a1) {
  asm volatile(
      "ds_read_b128 %0, %2\n\t"
      "ds_read_b128 %1, %3\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1])
      : "v"(a0), "v"(a1)
      : "memory");
}
__device__ __forceinline__ void lds_read2x16(f32x4_t (&o)[2], uint32_t addr) {
  asm volatile(
      "ds_read_b128 %0, %2\n\t"
      "ds_read_b128 %1, %2 offset:64\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1])
      : "v"(addr)
      : "memory");
}
// 16 B from LDS, read and waited for inside the statement
__device__ __forceinline__ f32x4_t lds_read1(uint32_t a0) {
  f32x4_t o;
  asm volatile(
      "ds_read_b128 %0, %1\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o)
      : "v"(a0)
      : "memory");
  return o;
}
__device__ __forceinline__ void lds_write16(uint32_t addr, f32x4_t v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write8(uint32_t addr, u32x2_t v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// the wave's earlier ds_writes, then 4 x 16 B at a0, a0 + 1024, a1, a1 + 1024, waited for in the statement
__device__ __forceinline__ void lds_read_rows(u32x4_t (&o)[4], uint32_t a0, uint32_t a1) {
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
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(size_t)(lptr_t)(p); }
typedef float f32x2_vt __attribute__((ext_vector_type(2)));
// 8 x 8 B at addr + 1024 i (one per wave of the workgroup), read and waited for inside the statement
__device__ __forceinline__ void lds_read8x8(f32x2_vt (&o)[8], uint32_t addr) {
  asm volatile(
      "ds_read_b64 %0, %8\n\t"
      "ds_read_b64 %1, %8 offset:1024\n\t"
      "ds_read_b64 %2, %8 offset:2048\n\t"
      "ds_read_b64 %3, %8 offset:3072\n\t"
      "ds_read_b64 %4, %8 offset:4096\n\t"
      "ds_read_b64 %5, %8 offset:5120\n\t"
      "ds_read_b64 %6, %8 offset:6144\n\t"
      "ds_read_b64 %7, %8 offset:7168\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(addr)
      : "memory");
}
// 8 x 8 B at addr + 32 i (the final statistics of rows r, r + 4, ..., r + 28)
__device__ __forceinline__ void lds_read8x8_s32(f32x2_vt (&o)[8], uint32_t addr) {
  asm volatile(
      "ds_read_b64 %0, %8\n\t"
      "ds_read_b64 %1, %8 offset:32\n\t"
      "ds_read_b64 %2, %8 offset:64\n\t"
      "ds_read_b64 %3, %8 offset:96\n\t"
      "ds_read_b64 %4, %8 offset:128\n\t"
      "ds_read_b64 %5, %8 offset:160\n\t"
      "ds_read_b64 %6, %8 offset:192\n\t"
      "ds_read_b64 %7, %8 offset:224\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(addr)
      : "memory");
}
// 8 x 8 B at addr + 256 i (the segment statistics of one row of a 32-row block, one per wave)
__device__ __forceinline__ void lds_read8x8_s256(f32x2_vt (&o)[8], uint32_t addr) {
  asm volatile(
      "ds_read_b64 %0, %8\n\t"
      "ds_read_b64 %1, %8 offset:256\n\t"
      "ds_read_b64 %2, %8 offset:512\n\t"
      "ds_read_b64 %3, %8 offset:768\n\t"
      "ds_read_b64 %4, %8 offset:1024\n\t"
      "ds_read_b64 %5, %8 offset:1280\n\t"
      "ds_read_b64 %6, %8 offset:1536\n\t"
      "ds_read_b64 %7, %8 offset:1792\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(addr)
      : "memory");
}
// x of this lane + x of lane ^ 32 (the same bits in both lanes): one v_permlane32_swap, no LDS
__device__ __forceinline__ float half_pair_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// x summed over the four lanes l, l ^ 16, l ^ 32, l ^ 48 as (x[l4 = 0] + x[2]) + (x[1] + x[3]), the same bits in all four
__device__ __forceinline__ float quad_lane_sum(float x) {
  const float t = half_pair_sum(x);
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int EPI, bool CONV, typename CFG, bool ROPE_LDS>
__global__ __launch_bounds__(512) void gemm_p_kernel(GemmParams p, GemmSeg sg, const bf16_t* zero16, int ntm, int ntn,
                                                     int grouped, int flags, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // NST x (A slot | B slot) | aux tables
  constexpr int WM = CFG::WM, WN = CFG::WN, MT = CFG::MT, NT = CFG::NT, BK = CFG::BK, NST = CFG::NST;
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  constexpr int ROWB = BK * 2, CH = ROWB / 16, RPI = 64 / CH, KK = BK / 16;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PW = A_BYTES / 8192, B_PW = B_BYTES / 8192, OPS = A_PW + B_PW, D = NST - 1;
  constexpr int PH = CFG::PH, KP = KK / PH;
  constexpr bool ROW_CFG_ANY = is_row_cfg<CFG>::value;
  constexpr bool RLN = (EPI == EPI_GATE_RES_LN);
  constexpr bool RES = (EPI == EPI_GATE_RES || EPI == EPI_RES || RLN);
  constexpr bool F32OUT = (EPI == EPI_BIAS_F32 || EPI == EPI_ACC_F32);
  // fp32 output on the 256x256 tile goes out like the fp32 residual: staged through LDS, whole lines (N % 256 == 0)
  constexpr bool F32ST = F32OUT && MT * NT * 4 > 16;
  constexpr bool CAN_EXACT = !F32OUT || F32ST;   // a full-tile epilogue leaves exactly E_ST stores in flight
  // store instructions per wave in a full-tile epilogue: bf16 tiles go out as whole 128-B lines (4 per 32 rows)
  // PRE: the fp32 residual tile fits in 16 float4 registers and is prefetched one K-step ahead (CfgRes);
  // otherwise (CfgBig) the residual is loaded inside the epilogue, block by block, in whole lines
  constexpr bool PRE = RES && MT * NT * 4 <= 16;
  constexpr bool STAGED32 = (RES && !PRE) || F32ST;
  constexpr bool HAS_X = RES || EPI == EPI_ACC_F32;   // the epilogue adds what the output location holds / xin
  constexpr bool ROW = is_row_cfg<CFG>::value;        // no aux tables in LDS; bf16 staging halves both in the B pieces
  constexpr bool LNG = (EPI == EPI_LN_GELU);
  static_assert(!(LNG || RLN) || (ROW && MT == 4), "row statistics need the whole row in one workgroup");
  constexpr bool ROWSTAT = LNG || RLN;   // the epilogue exchanges row statistics between the waves (barriers of its own)
  constexpr int E_ST = STAGED32 ? MT * 8 : (PRE || F32OUT) ? MT * NT * 4 : MT * 4;
  static_assert(WM * WN == 8 && A_PW * 8192 == A_BYTES && B_PW * 8192 == B_BYTES, "tile / wave layout");
  static_assert(NT == 2 && (A_PW >= 2 || ROW_CFG_ANY) && B_PW >= 2, "bf16 output staging: 32 x 64 block = two 2-KiB halves");
  static_assert(OPS * (D - 1) + E_ST < 64, "vmcnt range");
  static_assert(!STAGED32 || (A_PW >= 4 && B_PW >= 4) || ROW, "fp32 output staging: 32 x 64 block = two 4-KiB halves");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int nk = p.K / BK;
  float* auxf = reinterpret_cast<float*>(smem + NST * STAGE);
  const bool exact = CAN_EXACT && (flags & PF_EXACT) && nk > D;

  // ---- this workgroup's tile list
  int t_first, t_stride, n_my, xcd = 0;
  // CONV: the taps of a dilated convolution read the rows of the neighbouring panels (shift up to +-2 * dilation), so an
  // XCD owns a CONTIGUOUS run of panels and its CUs walk neighbouring panels at the same time -- the shifted rows are
  // then found in that XCD's L2 instead of being fetched from HBM once per tap (5x the operand bytes, PMC)
  const int ppx = (ntm + 7) >> 3;   // panels per XCD in the contiguous deal
  if (grouped) {  // gridDim.x == #CU: XCD group x owns panels x, x+8, ...  (CONV: panels x*ppx ... x*ppx + ppx - 1)
    xcd = blockIdx.x & 7;
    const int loc = blockIdx.x >> 3, gx = gridDim.x >> 3;
    int panels = (ntm - xcd + 7) >> 3;
    if constexpr (CONV) { panels = ntm - xcd * ppx; panels = panels < 0 ? 0 : (panels > ppx ? ppx : panels); }
    const int tiles_x = panels * ntn;
    t_first = loc; t_stride = gx;
    n_my = tiles_x > loc ? (tiles_x - loc + gx - 1) / gx : 0;
  } else {
    t_first = blockIdx.x; t_stride = gridDim.x;
    const int tot = ntm * ntn;
    n_my = tot > t_first ? (tot - t_first + t_stride - 1) / t_stride : 0;
  }
  if (n_my == 0) return;
#ifdef SGPO_GEMM_ABLATE
  // Diagnostic build only: half of the workgroups of every XCD start `dbg >> 8` microseconds late, so that their
  // HBM-bound epilogues would fall into the other half's MFMA-bound K loops instead of all at once.  Measured (round
  // 2): no gain at 15 / 30 / 45 us on any of the three GEMMs -- the start offset does not survive.
  if ((dbg >> 8) && (blockIdx.x & 8)) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), ticks = (unsigned long long)(dbg >> 8) * 100ull;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  }
#endif
  auto tile_origin = [&](int i, int& m0, int& n0) {
    const int j = t_first + i * t_stride;
    if (grouped) { m0 = (CONV ? xcd * ppx + j / ntn : (j / ntn) * 8 + xcd) * BM; n0 = (j % ntn) * BN; }
    else { m0 = (j / ntn) * BM; n0 = (j % ntn) * BN; }
  };

  // ---- aux tables -> LDS (once per launch; the only ordinary loads outside RES / fp32-out epilogues)
  if constexpr ((!F32OUT || F32ST) && !ROW) {
    if constexpr (EPI == EPI_QKV_ROPE) {
      if constexpr (ROPE_LDS) {
        const int n = p.L * (p.head_dim / 2);
        // rows of 32 floats (head_dim 64); lanes read rows of consecutive positions at the same column,
        // so the 16-B chunk index is XORed with (pos >> 1) & 7: conflict-free for ds_read_b128
        for (int i = tid; i < n; i += 512) {
          const int pos = i >> 5, j = ((((i >> 2) & 7) ^ ((pos >> 1) & 7)) << 2) | (i & 3);
          auxf[pos * 32 + j] = p.rope_cos[i];
          auxf[n + pos * 32 + j] = p.rope_sin[i];
        }
      }
    } else {
      for (int i = tid; i < p.N; i += 512) auxf[i] = p.bias[i];
    }
    __syncthreads();
  }

  // ---- LDS-DMA staging.  The stream position being staged is tracked incrementally (st_i, st_kt).
  int st_i = 0, st_kt = 0, st_m0, st_n0;
  tile_origin(0, st_m0, st_n0);
  // per-lane byte offsets of the DMA pieces inside a full tile (row * ld + swizzled chunk): constant for
  // the whole launch, so staging a full tile costs one scalar base per operand + OPS DMA instructions
  // (pieces j and j + 2 of an operand are 16 rows apart and share the swizzle term, so two offsets per operand and a
  //  scalar row stride describe them all)
  static_assert(CH == 8 && (A_PW % 2 == 0 || A_PW == 1) && B_PW % 2 == 0, "piece offsets: 128-B rows, pairs of pieces");
  uint32_t a_off[2], w_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave * A_PW + j) * RPI + lane / CH;
    const int c = (lane % CH) ^ ((row >> 1) & 7);
    a_off[j] = ((uint32_t)row * (uint32_t)p.lda + c * 8) * 2;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave * B_PW + j) * RPI + lane / CH;
    const int c = (lane % CH) ^ ((row >> 1) & 7);
    w_off[j] = ((uint32_t)row * (uint32_t)p.ldw + c * 8) * 2;
  }
  const size_t a_pair = (size_t)(2 * RPI) * p.lda * 2, w_pair = (size_t)(2 * RPI) * p.ldw * 2;   // bytes between pieces j, j + 2
  auto stage = [&](int slot) {
    const int kglob = st_kt * BK;
    char* abase = smem + slot * STAGE;
    char* bbase = abase + A_BYTES;
    if (!CONV && st_m0 + BM <= p.M && st_n0 + BN <= p.N) {
      const char* ag = reinterpret_cast<const char*>(p.A + (size_t)st_m0 * p.lda + kglob);
      const char* wg = reinterpret_cast<const char*>(p.W + (size_t)st_n0 * p.ldw + kglob);
#pragma unroll
      for (int j = 0; j < A_PW; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(ag + (j >> 1) * a_pair + a_off[j & 1]), (lptr_t)(abase + (wave * A_PW + j) * 1024), 16, 0, 0);
#pragma unroll
      for (int j = 0; j < B_PW; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(wg + (j >> 1) * w_pair + w_off[j & 1]), (lptr_t)(bbase + (wave * B_PW + j) * 1024), 16, 0, 0);
    } else {
      int kin = kglob, sh = 0;
      if constexpr (CONV) {
        const int seg = kglob / sg.seg_k;
        kin = kglob - seg * sg.seg_k;
        sh = sg.shift[seg];
      }
#pragma unroll
      for (int j = 0; j < A_PW; ++j) {
        const int ii = wave * A_PW + j;
        const int row = ii * RPI + lane / CH;
        const int c = (lane % CH) ^ (CH == 8 ? (row >> 1) & 7 : (row >> 2) & 3);
        int m = st_m0 + row;
        if (m > p.M - 1) m = p.M - 1;
        const bf16_t* src = p.A + ((size_t)m + sh) * p.lda + kin + c * 8;
        if constexpr (CONV) {
          const int lp = m % p.L + sh;
          if (lp < 0 || lp >= p.L) src = zero16;
        }
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(abase + ii * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < B_PW; ++j) {
        const int ii = wave * B_PW + j;
        const int row = ii * RPI + lane / CH;
        const int c = (lane % CH) ^ (CH == 8 ? (row >> 1) & 7 : (row >> 2) & 3);
        int n = st_n0 + row;
        if (n > p.N - 1) n = p.N - 1;
        __builtin_amdgcn_global_load_lds((gptr_t)(p.W + (size_t)n * p.ldw + kglob + c * 8), (lptr_t)(bbase + ii * 1024), 16, 0, 0);
      }
    }
    if (++st_kt == nk) {
      st_kt = 0;
      if (++st_i < n_my) tile_origin(st_i, st_m0, st_n0);
    }
  };

  // ---- fragment addressing inside a slot.  The products are v_mfma_f32_16x16x32_bf16 (MI355X_MICROARCH.md, DVFS item 7:
  // the chip holds a higher clock on this shape than on 32x32x16 at equal cycles per flop; the K loops measured 6-8 % shorter,
  // profiles/r3/r3f_gemm_kloop_pricing.txt).  A phase is one 32-wide K slice: lane (l16 = lane & 15, l4 = lane >> 4) reads
  // the 8 consecutive k of chunk ph*4 + l4 of row l16 of a 16-row group -- one ds_read_b128 per 16 rows, the same count
  // and the same bank pattern (16 consecutive rows per 16 lanes) as the 32-row fragments of the other shape.
  static_assert(CH == 8 && KK == 4 && PH == 2, "one 32-wide K slice per phase, 128-B rows");
  const int l16 = lane & 15, l4 = lane >> 4;
  const int a_lane = (wm * MT * 32 + l16) * ROWB;
  const int b_lane = A_BYTES + (wn * NT * 32 + l16) * ROWB;
  int cx[PH];   // rows 16 h + l16 of any 32-row block share the swizzle term
#pragma unroll
  for (int ph = 0; ph < PH; ++ph) cx[ph] = ((ph * 4 + l4) ^ ((l16 >> 1) & 7)) << 4;

  // A 32x32 output block is four 16x16 products: acc[nt][mt][2 nh + mh][q] = D[n local = nt*32 + 16 nh + 4 l4 + q][m local =
  // mt*32 + 16 mh + l16] -- a lane owns TWO rows per block and two column quads of each
  f32x4_t acc[NT][MT][4];
  float4 res[PRE ? MT * NT * 4 : 1];

  // ---- epilogue of tile (m0, n0): lane owns rows m(mt, mh) and column quads n = nb0 + nt*32 + 16 nh + 4 l4
  // bf16 output block of one mt (32 rows x 64 columns, 4 KiB) goes through LDS so that every store
  // instruction writes 8 whole 128-B lines (lane = 16 B) instead of 8 B in each of 32 lines: the
  // row-per-lane store tail is bound by store instructions x lines touched, not by bytes.  The block
  // lives in the pieces of the FREE ring slot that this wave itself fills next (its own LDS-DMA
  // destinations), so no other wave touches it and no barrier is needed.  Image: row r at
  // (r & 15) * 128 in half r >> 4; 16-B chunk ch at ch ^ (r & 7); the 8-B halves of a chunk are
  // swapped when r & 8 (conflict-free ds_write_b64 for the 16-lane groups).
  auto blk_h0 = [&](int fslot) {
    char* fs = smem + fslot * STAGE;
    return ROW ? lds_addr(fs + A_BYTES + (wave * B_PW) * 1024) : lds_addr(fs + (wave * A_PW) * 1024);
  };
  auto blk_h1 = [&](int fslot, uint32_t h0) {
    return (ROW || A_PW >= 4) ? h0 + 2048 : lds_addr(smem + fslot * STAGE + A_BYTES + (wave * B_PW) * 1024);
  };
  // one column quad (nt, g) of this lane's row into the block image
  auto block_put = [&](uint32_t h0, uint32_t h1, int nt, int g, u32x2_t v, int lr, int lh) {
    const uint32_t wbase = ((lr >> 4) ? h1 : h0) + (lr & 15) * 128 + ((lh ^ ((lr >> 3) & 1)) << 3);
    lds_write8(wbase + (((nt * 4 + g) ^ (lr & 7)) << 4), v);
  };
  // the block image back as whole lines, to global memory
  auto block_flush = [&](uint32_t h0, uint32_t h1, int m_blk, int nb0, int le, auto full_tag) {   // le = lane (the epilogue's opaque copy)
    constexpr bool FULL = decltype(full_tag)::value;
    const int r8 = le >> 3, c8 = le & 7;
    const uint32_t roff = r8 * 128 + ((c8 ^ r8) << 4);
    u32x4_t o[4];
    lds_read_rows(o, h0 + roff, h1 + roff);
    bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + (size_t)(m_blk + r8) * p.ldc + nb0 + c8 * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4_t v = o[i];
      if (i & 1) v = u32x4_t{o[i].z, o[i].w, o[i].x, o[i].y};   // rows with bit 3 set hold swapped halves
      if (ABL(16)) { asm volatile("" ::"v"(v)); continue; }   // diagnostic: no output stores
      if (ABL(128)) { if (FULL || m_blk + i * 8 + r8 < p.M) *reinterpret_cast<u32x4_t*>(cp + (size_t)(i * 8) * p.ldc) = v; continue; }   // diagnostic: plain stores
      if (FULL || m_blk + i * 8 + r8 < p.M)
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(cp + (size_t)(i * 8) * p.ldc));   // streamed once: keep L2 for the operand panels
    }
  };
  // fp32 residual epilogue of the 256x256 tile, one mt block (32 rows x 64 columns, 8 KiB) at a time:
  // y = gate * (acc + bias) goes through the wave's own pieces of the free slot (as the bf16 blocks do) and
  // comes back one 16-B chunk per lane, 4 whole 256-B rows per instruction; the residual x is loaded
  // and x + y stored in that layout, so every global access is a run of whole lines.  Image: row r at
  // (r & 15) * 256 in half r >> 4, 16-B chunk c at c ^ (r & 7).
  auto res_block = [&](int m_blk, int nb0, uint32_t h0, uint32_t h1, int r4, int c16, auto full_tag) {   // r4 = lane >> 4, c16 = lane & 15
    constexpr bool FULL = decltype(full_tag)::value;
    float* cp = reinterpret_cast<float*>(p.C) + (size_t)(m_blk + r4) * p.ldc + nb0 + c16 * 4;
    const float* xp = (RES ? p.xin : reinterpret_cast<const float*>(p.C)) + (size_t)(m_blk + r4) * p.ldc + nb0 + c16 * 4;
    f32x4_t xr[HAS_X ? 8 : 1];
    if constexpr (HAS_X) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        size_t ro = (size_t)(i * 4) * p.ldc;
        if (!FULL && m_blk + i * 4 + r4 > p.M - 1) ro = 0;   // any valid row: the value is not stored
        if (ABL(32)) xr[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};   // diagnostic: no residual loads
        else xr[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(xp + ro));
      }
    }
    const uint32_t k0 = r4 * 256 + ((c16 ^ r4) << 4), k1 = r4 * 256 + ((c16 ^ (r4 + 4)) << 4);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const uint32_t hb = hf ? h1 : h0;
      f32x4_t o[4];
      lds_read4(o, hb + k0, hb + 1024 + k1, hb + 2048 + k0, hb + 3072 + k1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = hf * 4 + j;
        f32x4_t v = o[j];
        if constexpr (HAS_X) {
          const f32x4_t x4 = xr[i];
          v = f32x4_t{rn_add(x4.x, o[j].x), rn_add(x4.y, o[j].y), rn_add(x4.z, o[j].z), rn_add(x4.w, o[j].w)};
        }
        if (ABL(16)) { asm volatile("" ::"v"(v)); continue; }   // diagnostic: no output stores
        if (ABL(128)) { if (FULL || m_blk + i * 4 + r4 < p.M) *reinterpret_cast<f32x4_t*>(cp + (size_t)(i * 4) * p.ldc) = v; continue; }   // diagnostic: plain stores
        if (FULL || m_blk + i * 4 + r4 < p.M) __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(cp + (size_t)(i * 4) * p.ldc));
      }
    }
  };

  auto epilogue = [&](int m0, int n0, int fslot, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const int nb0 = n0 + wn * NT * 32;
    if (!FULL && !(F32OUT && !F32ST) && nb0 >= p.N) return;   // wave-uniform: this wave's 64 columns lie outside N
    // this lane's place in the 16x16 products (rows 16 mh + e16 of a 32-row block, column quads 16 nh + 4 e4 of a 32-column
    // block), formed from an opaque copy INSIDE the epilogue: derived from `lane` directly the values are tile-invariant,
    // get hoisted out of the persistent loop and stay live through the K loop
    int le = lane;
    asm volatile("" : "+v"(le));
    const int e16 = le & 15, e4 = le >> 4, ec = e4 >> 1, eh = e4 & 1;
    if constexpr (EPI == EPI_QKV_ROPE) {
      const bool rope = nb0 < p.rope_cols;  // wave-uniform: the 64 columns are one head of q, k or v
      const int hd2 = p.head_dim / 2;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m_blk = m0 + wm * MT * 32 + mt * 32;
        const uint32_t h0 = blk_h0(fslot), h1 = blk_h1(fslot, h0);
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
          const int row = mh * 16 + e16;
          const int m = m_blk + row;
          const int pos = ((FULL || m < p.M) ? m : 0) % p.L;
          f32x4_t cs_l[2], sn_l[2];   // table chunks 4 nh + e4 of this position (columns d = 16 nh + 4 e4 of the head's first half)
          if constexpr (ROPE_LDS) {
            const uint32_t rb = lds_addr(auxf + pos * 32), key = (pos >> 1) & 7, sb = (uint32_t)p.L * 128;
            const uint32_t c0 = ((uint32_t)e4 ^ key) << 4, c1 = ((uint32_t)(e4 + 4) ^ key) << 4;
            lds_read2(cs_l, rb + c0, rb + c1);
            lds_read2(sn_l, rb + sb + c0, rb + sb + c1);
          }
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) {
            const int d = 16 * nh + 4 * e4;
            const f32x4_t x0 = acc[0][mt][2 * nh + mh], x1 = acc[1][mt][2 * nh + mh];
            float a0[4] = {x0.x, x0.y, x0.z, x0.w}, a1[4] = {x1.x, x1.y, x1.z, x1.w};
            if (rope) {
              f32x4_t cs, sn;
              if constexpr (ROPE_LDS) {
                cs = cs_l[nh]; sn = sn_l[nh];
              } else {
                cs = *reinterpret_cast<const f32x4_t*>(p.rope_cos + pos * hd2 + d);
                sn = *reinterpret_cast<const f32x4_t*>(p.rope_sin + pos * hd2 + d);
              }
              const float c4[4] = {cs.x, cs.y, cs.z, cs.w}, s4[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float o0, o1;
                epi_rope(a0[q], a1[q], c4[q], s4[q], o0, o1);
                a0[q] = o0; a1[q] = o1;
              }
            }
            block_put(h0, h1, 0, 2 * nh + ec, pack4(a0[0], a0[1], a0[2], a0[3]), row, eh);
            block_put(h0, h1, 1, 2 * nh + ec, pack4(a1[0], a1[1], a1[2], a1[3]), row, eh);
          }
        }
        block_flush(h0, h1, m_blk, nb0, le, full_tag);
        __builtin_amdgcn_sched_barrier(0);   // one block at a time: keeps the epilogue's register peak down
      }
    } else if constexpr (F32OUT && !F32ST) {
      float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
          const int m = m0 + wm * MT * 32 + mt * 32 + mh * 16 + e16;
          const bool mok = m < p.M;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
              const int n = nb0 + nt * 32 + 16 * nh + 4 * e4;
#pragma unroll
              for (int q = 0; q < 4; ++q)
                if (mok && n + q < p.N) {
                  float o = acc[nt][mt][2 * nh + mh][q] + p.bias[n + q];
                  if constexpr (EPI == EPI_ACC_F32) o += C[(size_t)m * p.ldc + n + q];
                  C[(size_t)m * p.ldc + n + q] = o;
                }
            }
        }
    } else if constexpr (RLN) {
      // x_new = x + gate (acc + bias) as in the staged fp32 residual epilogue (same arithmetic, same bits), one 32-row block
      // at a time: y goes through the wave's own pieces of the free slot and comes back transposed (a lane holds the quad
      // c16 of rows i*4 + r4), x is loaded and x_new stored in that layout and KEPT in registers.  Per row and wave one
      // segment statistic (ln_stats.h) goes to the A region of the free slot, [wave][row of the block], double-buffered by
      // block parity; after ONE barrier every wave combines the eight segments of the block's 32 rows (lane = row), hands
      // (mean, rstd) to the lanes that hold the row's quads (ds_bpermute) and stores the modulated bf16 row pieces (16 lanes
      // = one whole 128-B line).  A last barrier keeps the next prefetch from overwriting statistics still being read.
      char* fs = smem + fslot * STAGE;
      const uint32_t st_base = lds_addr(fs);          // 2 x [wave][32 rows] (mean_s, M2_s): 4 KiB
      const uint32_t h0 = lds_addr(fs + A_BYTES + (wave * B_PW) * 1024), h1 = h0 + 4096;
      const int lr = le & 31;
      const int r4 = e4, c16 = e16;
      const uint32_t k0 = r4 * 256 + ((c16 ^ r4) << 4), k1 = r4 * 256 + ((c16 ^ (r4 + 4)) << 4);
      const bool per_seq = p.gate_stride != 0;        // wave-uniform: one modulation row per sequence, else one for all
      const int nq = nb0 + c16 * 4;
      // The per-column tables of this wave's 64 columns (bias | gate | ln_w | ln_scale | ln_shift, 5 x 256 B) go to LDS once per
      // tile, behind the statistics in the A region of the free slot (wave-private, no barrier).  Read from global memory
      // where they are used they were 11-33 vector-memory loads per 32-row block, and in the in-order VM queue each of
      // them waited for the previous block's stores to be acknowledged (timing-only build without them: 1098 -> 966 us).
      // With one modulation row per sequence only bias and ln_w are taken from here.
      const uint32_t tb = st_base + 4096 + wave * 1280;
      {
        const int tsel = le >> 4;
        const float* src = tsel == 0 ? p.bias : tsel == 1 ? p.gate : tsel == 2 ? p.ln_w : p.ln_scale;
        lds_write16(tb + le * 16, *reinterpret_cast<const f32x4_t*>(src + nb0 + 4 * e16));
        if (le < 16) lds_write16(tb + 1024 + le * 16, *reinterpret_cast<const f32x4_t*>(p.ln_shift + nb0 + 4 * le));
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m_blk = m0 + mt * 32;
        const uint32_t area = st_base + (mt & 1) * 2048;
        // (table loads are vector-memory instructions like the residual traffic and share its pipe: the bias quad is loaded
        //  once for both rows, and so is the gate quad when all sequences share one modulation row)
        size_t goff[2] = {0, 0};
        if (per_seq) {
#pragma unroll
          for (int mh = 0; mh < 2; ++mh) {
            const int row = mh * 16 + e16;
            int b = ((FULL || m_blk + row < p.M) ? m_blk + row : p.M - 1) / p.L;
            if (b > p.B - 1) b = p.B - 1;
            goff[mh] = (size_t)b * p.gate_stride;
          }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) {
            int n = nb0 + nt * 32 + 16 * nh + 4 * e4;
            asm volatile("" : "+v"(n));   // opaque: the table loads of the blocks are not merged and kept live
            const uint32_t tq = tb + (uint32_t)(n - nb0) * 4;
            f32x4_t bg[2], g4[2];
            lds_read2(bg, tq, tq + 256);
            const f32x4_t b4 = bg[0];
            g4[0] = bg[1];
            g4[1] = bg[1];
            if (per_seq) {
              g4[0] = *reinterpret_cast<const f32x4_t*>(p.gate + goff[0] + n);
              g4[1] = *reinterpret_cast<const f32x4_t*>(p.gate + goff[1] + n);
            }
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
              const uint32_t wb = (mh ? h1 : h0) + e16 * 256;
              const f32x4_t a = acc[nt][mt][2 * nh + mh];
              const f32x4_t v = {rn_mul(a.x + b4.x, g4[mh].x), rn_mul(a.y + b4.y, g4[mh].y), rn_mul(a.z + b4.z, g4[mh].z),
                                 rn_mul(a.w + b4.w, g4[mh].w)};
              lds_write16(wb + (((nt * 8 + 4 * nh + e4) ^ (e16 & 7)) << 4), v);
            }
          }
        float* cp = reinterpret_cast<float*>(p.C) + (size_t)(m_blk + r4) * p.ldc + nq;
        const float* xp = p.xin + (size_t)(m_blk + r4) * p.ldc + nq;
        f32x4_t xs[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          size_t ro = (size_t)(i * 4) * p.ldc;
          if (!FULL && m_blk + i * 4 + r4 > p.M - 1) ro = 0;   // any valid row: the value is not stored
          xs[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(xp + ro));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const uint32_t hb = hf ? h1 : h0;
          f32x4_t o[4];
          lds_read4(o, hb + k0, hb + 1024 + k1, hb + 2048 + k0, hb + 3072 + k1);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = hf * 4 + j;
            const f32x4_t x4 = xs[i];
            const f32x4_t v = {rn_add(x4.x, o[j].x), rn_add(x4.y, o[j].y), rn_add(x4.z, o[j].z), rn_add(x4.w, o[j].w)};
            xs[i] = v;
            if (FULL || m_blk + i * 4 + r4 < p.M) __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(cp + (size_t)(i * 4) * p.ldc));
            float m_s, q_s;
            lns_segment(v.x, v.y, v.z, v.w, m_s, q_s);
            u32x2_t w2;
            w2.x = __float_as_uint(m_s); w2.y = __float_as_uint(q_s);
            if (c16 == 0) lds_write8(area + ((wave * 32 + i * 4 + r4) << 3), w2);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float mean_r, rstd_r;   // of row lr of the block (both halves of the wave compute the same 32 rows)
        {
          f32x2_vt pw[8];
          lds_read8x8_s256(pw, area + (lr << 3));
          float ms[8], qs[8];
#pragma unroll
          for (int s8 = 0; s8 < 8; ++s8) { ms[s8] = pw[s8].x; qs[s8] = pw[s8].y; }
          lns_combine8(ms, qs, p.ln_eps, mean_r, rstd_r);
        }
        const f32x4_t w4 = lds_read1(tb + 512 + c16 * 16);
        bf16_t* op = p.ln_out + (size_t)(m_blk + r4) * p.N + nq;
        f32x4_t cs[2];   // one modulation row for all sequences: read once per block (per row below otherwise)
        lds_read2(cs, tb + 768 + c16 * 16, tb + 1024 + c16 * 16);
        f32x4_t c4 = cs[0], s4 = cs[1];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int m = m_blk + i * 4 + r4;
          const float mean = __int_as_float(__builtin_amdgcn_ds_bpermute((i * 4 + r4) << 2, __float_as_int(mean_r)));
          const float rstd = __int_as_float(__builtin_amdgcn_ds_bpermute((i * 4 + r4) << 2, __float_as_int(rstd_r)));
          if (per_seq) {
            int b = ((FULL || m < p.M) ? m : p.M - 1) / p.L;
            if (b > p.B - 1) b = p.B - 1;
            const size_t moff = (size_t)b * p.gate_stride;
            c4 = *reinterpret_cast<const f32x4_t*>(p.ln_scale + moff + nq);
            s4 = *reinterpret_cast<const f32x4_t*>(p.ln_shift + moff + nq);
          }
          const f32x4_t v = xs[i];
          u32x2_t pk;
          pk.x = pack_bf16x2(lns_modulate(v.x, mean, rstd, w4.x, c4.x, s4.x), lns_modulate(v.y, mean, rstd, w4.y, c4.y, s4.y));
          pk.y = pack_bf16x2(lns_modulate(v.z, mean, rstd, w4.z, c4.z, s4.z), lns_modulate(v.w, mean, rstd, w4.w, c4.w, s4.w));
          if (FULL || m < p.M) *reinterpret_cast<u32x2_t*>(op + (size_t)(i * 4) * p.N) = pk;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_barrier();   // every wave has read its last statistics: the slot may be refilled
    } else if constexpr (LNG) {
      // bias, then LayerNorm over the 512 outputs of each row, affine, exact GELU, bf16.  A lane holds 16 values of each of
      // its 8 rows (mt, mh): the quads at columns 32 nt + 16 nh + 4 e4 of the wave's 64; lanes e4 = 0..3 hold the rest of the
      // segment, the other seven waves the rest of the row.  Each wave reduces its 64 columns to (mean_w, M2_w = sum of
      // squared deviations from mean_w): per lane the quad sums accumulated over (nt, nh), then (e4: (0 + 2) + (1 + 3)); the
      // eight partials are combined exactly (Chan et al.): mean = avg(mean_w), M2 = sum M2_w + 64 sum (mean_w - mean)^2 -- as
      // accurate as the two-pass form, one exchange.  Partials live in the A region of the free slot ([wave][mt][row] float2,
      // 8 KiB): written, barrier, read by every wave, barrier (the next prefetch overwrites it).  Both wave groups run
      // their epilogue in the same barrier interval, so the two extra barriers pair up.  ln_gelu_row512_kernel (bytenet.hip)
      // repeats this arithmetic operation for operation.
      const uint32_t st_base = lds_addr(smem + fslot * STAGE);
      // (the accumulators are only read: acc + bias is formed again in each of the three passes -- updating them in place
      //  made the register allocator keep a second copy and spill.)  The tables of this wave's 64 columns (bias | ln_w |
      // ln_b, 3 x 256 B) go to LDS once per tile, behind the partials (wave-private): as vector-memory loads at their
      // points of use each waited, in the in-order VM queue, for the stores of the block before it
      const uint32_t tb = st_base + 8192 + wave * 768;
      if (le < 48) {
        const int tsel = le >> 4;
        const float* src = tsel == 0 ? p.bias : tsel == 1 ? p.ln_w : p.ln_b;
        lds_write16(tb + le * 16, *reinterpret_cast<const f32x4_t*>(src + nb0 + 4 * e16));
      }
      float mean[MT][2], rstd[MT][2];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) { mean[mt][0] = 0.f; mean[mt][1] = 0.f; }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          int n = nb0 + nt * 32 + 16 * nh + 4 * e4;
          asm volatile("" : "+v"(n));   // opaque: identical loads of the three passes are not merged (and kept live)
          const f32x4_t b4 = lds_read1(tb + (uint32_t)(n - nb0) * 4);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
              const f32x4_t a = acc[nt][mt][2 * nh + mh];
              mean[mt][mh] += ((a.x + b4.x) + (a.y + b4.y)) + ((a.z + b4.z) + (a.w + b4.w));
            }
        }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) { mean[mt][mh] = quad_lane_sum(mean[mt][mh]) * (1.0f / 64.0f); rstd[mt][mh] = 0.f; }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          int n = nb0 + nt * 32 + 16 * nh + 4 * e4;
          asm volatile("" : "+v"(n));
          const f32x4_t b4 = lds_read1(tb + (uint32_t)(n - nb0) * 4);
          const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float d = (acc[nt][mt][2 * nh + mh][q] + bv[q]) - mean[mt][mh];
                rstd[mt][mh] = fmaf(d, d, rstd[mt][mh]);
              }
        }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
          const float sq = quad_lane_sum(rstd[mt][mh]);
          if (e4 == 0) {
            u32x2_t w2;
            w2.x = __float_as_uint(mean[mt][mh]); w2.y = __float_as_uint(sq);
            lds_write8(st_base + (((wave * MT + mt) * 32 + mh * 16 + e16) << 3), w2);
          }
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // (the eight partials of a row are combined where the row is normalised, block by block: only two rows' (mean, rstd)
      //  are live at a time instead of eight)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m_blk = m0 + mt * 32;
        const uint32_t h0 = blk_h0(fslot), h1 = blk_h1(fslot, h0);
        float mu2[2], rs2[2];
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
          f32x2_vt pw[8];
          lds_read8x8(pw, st_base + ((mt * 32 + mh * 16 + e16) << 3));
          float ms = 0.f, m2 = 0.f;
#pragma unroll
          for (int w = 0; w < 8; ++w) { ms += pw[w].x; m2 += pw[w].y; }
          const float mu = ms * 0.125f;
          float dv = 0.f;
#pragma unroll
          for (int w = 0; w < 8; ++w) { const float d = pw[w].x - mu; dv = fmaf(d, d, dv); }
          m2 = fmaf(64.0f, dv, m2);
          mu2[mh] = mu;
          rs2[mh] = rsqrtf(fmaf(m2, 1.0f / (float)BN, p.ln_eps));   // explicit: ln_gelu_row512_kernel (bytenet.hip) mirrors this
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) {
            int n = nb0 + nt * 32 + 16 * nh + 4 * e4;
            asm volatile("" : "+v"(n));
            const uint32_t tq = tb + (uint32_t)(n - nb0) * 4;
            f32x4_t bw[2];
            lds_read2(bw, tq, tq + 256);
            const f32x4_t b4 = bw[0], w4 = bw[1], c4 = lds_read1(tq + 512);
            const float bv[4] = {b4.x, b4.y, b4.z, b4.w}, wv[4] = {w4.x, w4.y, w4.z, w4.w}, cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
              float v[4];
#pragma unroll
              for (int q = 0; q < 4; ++q)
                v[q] = epi_gelu_erf_fast(fmaf(rn_mul((acc[nt][mt][2 * nh + mh][q] + bv[q]) - mu2[mh], rs2[mh]), wv[q], cv[q]));
              block_put(h0, h1, nt, 2 * nh + ec, pack4(v[0], v[1], v[2], v[3]), mh * 16 + e16, eh);
            }
          }
        block_flush(h0, h1, m_blk, nb0, le, full_tag);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_barrier();   // every wave has read its last partials: the slot may be refilled
      asm volatile("" ::: "memory");
    } else if constexpr (STAGED32) {
      char* fs = smem + fslot * STAGE;
      const uint32_t h0 = lds_addr(fs + (wave * A_PW) * 1024);
      const uint32_t h1 = lds_addr(fs + A_BYTES + (wave * B_PW) * 1024);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m_blk = m0 + wm * MT * 32 + mt * 32;
        f32x4_t bb[NT][2];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) lds_read2x16(bb[nt], lds_addr(auxf + nb0 + nt * 32 + 4 * e4));
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
          const int row = mh * 16 + e16;
          const uint32_t wb = (mh ? h1 : h0) + e16 * 256;
          int b = ((FULL || m_blk + row < p.M) ? m_blk + row : p.M - 1) / p.L;
          if (b > p.B - 1) b = p.B - 1;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
              const f32x4_t a = acc[nt][mt][2 * nh + mh], b4 = bb[nt][nh];
              f32x4_t v = {a.x + b4.x, a.y + b4.y, a.z + b4.z, a.w + b4.w};
              if constexpr (EPI == EPI_GATE_RES) {
                const f32x4_t g4 = *reinterpret_cast<const f32x4_t*>(p.gate + (size_t)b * p.gate_stride + nb0 + nt * 32 + 16 * nh + 4 * e4);
                v = f32x4_t{rn_mul(v.x, g4.x), rn_mul(v.y, g4.y), rn_mul(v.z, g4.z), rn_mul(v.w, g4.w)};
              }
              lds_write16(wb + (((nt * 8 + 4 * nh + e4) ^ (e16 & 7)) << 4), v);
            }
        }
        res_block(m_blk, nb0, h0, h1, e4, e16, full_tag);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      f32x4_t bb[NT][2];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) lds_read2x16(bb[nt], lds_addr(auxf + nb0 + nt * 32 + 4 * e4));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m_blk = m0 + wm * MT * 32 + mt * 32;
        uint32_t h0 = 0, h1 = 0;
        if constexpr (!RES) { h0 = blk_h0(fslot); h1 = blk_h1(fslot, h0); }
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
          const int row = mh * 16 + e16;
          const int m = m_blk + row;
          const bool mok = FULL || m < p.M;
          float4 gg[EPI == EPI_GATE_RES ? NT * 2 : 1];
          if constexpr (EPI == EPI_GATE_RES) {
            int b = m / p.L;
            if (b > p.B - 1) b = p.B - 1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int nh = 0; nh < 2; ++nh)
                gg[nt * 2 + nh] = *reinterpret_cast<const float4*>(p.gate + (size_t)b * p.gate_stride + nb0 + nt * 32 + 16 * nh + 4 * e4);
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
              const int n = nb0 + nt * 32 + 16 * nh + 4 * e4;
              const f32x4_t a = acc[nt][mt][2 * nh + mh], b4 = bb[nt][nh];
              float v[4] = {a.x + b4.x, a.y + b4.y, a.z + b4.z, a.w + b4.w};
              if constexpr (EPI == EPI_BIAS_GELU) if (!ABL(64)) {   // diagnostic bit 64: no GELU arithmetic
                const f32x2_t g0 = epi_gelu_tanh2(f32x2_t{v[0], v[1]}), g1 = epi_gelu_tanh2(f32x2_t{v[2], v[3]});
                v[0] = g0.x; v[1] = g0.y; v[2] = g1.x; v[3] = g1.y;
              }
              if constexpr (EPI == EPI_BIAS_GELU_ERF) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = epi_gelu_erf(v[q]);
              }
              if constexpr (RES) {
                if constexpr (EPI == EPI_GATE_RES) {   // v already holds acc + bias (epi_gate_mul / epi_gate_res)
                  const float4 g4 = gg[nt * 2 + nh];
                  v[0] = rn_mul(v[0], g4.x); v[1] = rn_mul(v[1], g4.y);
                  v[2] = rn_mul(v[2], g4.z); v[3] = rn_mul(v[3], g4.w);
                }
                if constexpr (PRE) {
                  if (mok) {
                    const float4 r = res[((mt * 2 + mh) * NT + nt) * 2 + nh];
                    const float4 o = make_float4(rn_add(r.x, v[0]), rn_add(r.y, v[1]), rn_add(r.z, v[2]), rn_add(r.w, v[3]));
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n) = o;
                  }
                }
              } else {
                block_put(h0, h1, nt, 2 * nh + ec, pack4(v[0], v[1], v[2], v[3]), row, eh);
              }
            }
        }
        if constexpr (!RES) block_flush(h0, h1, m_blk, nb0, le, full_tag);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  auto load_residual = [&](int m0, int n0) {
    if constexpr (PRE) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
          int m = m0 + wm * MT * 32 + mt * 32 + mh * 16 + l16;
          if (m > p.M - 1) m = p.M - 1;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
              const int n = n0 + wn * NT * 32 + nt * 32 + 16 * nh + 4 * l4;
              res[((mt * 2 + mh) * NT + nt) * 2 + nh] = *reinterpret_cast<const float4*>(p.xin + (size_t)m * p.ldc + n);
            }
        }
    }
  };

  auto tile_is_full = [&](int m0, int n0) { return m0 + BM <= p.M && n0 + BN <= p.N; };

  // wait until the stage `ahead` stages behind the newest one has landed; `extra` adds X = MT*NT*4
  // younger VM operations: the stores of a full-tile epilogue, or the residual prefetch
  auto wait_stage = [&](int ahead, bool extra) {
    if (extra) {
      if (ahead >= D - 1) wait_vm<OPS*(D - 1) + E_ST>();
      else if (D - 1 >= 2 && ahead == 1) wait_vm<OPS + E_ST>();   // only near the end of the stream
      else wait_vm<E_ST>();
      return;
    }
    if (ahead >= D - 1) wait_vm<OPS*(D - 1)>();
    else if (D - 1 >= 2 && ahead == 1) wait_vm<OPS>();
    else wait_vm<0>();
  };

  // ---- main stream.  Each K-step is PH phases; a phase is R (fragment reads; in the first phase
  // also epilogue, prefetch issue and the wait for the NEXT step's stage) | barrier | M (MFMAs) |
  // barrier.  Waves w and w+4 share a SIMD; waves 4-7 run one barrier behind waves 0-3, so on every
  // SIMD one wave's M phase overlaps the other's R phase (ping-pong).  With barriers numbered b0..:
  // group 0 has R_f before b_2f and M_f after it, group 1 has R_f after b_2f and M_f after b_2f+1.
  //   RAW: every wave waits for its own DMA of stage s+1 at the end of the last R of step s, i.e.
  //        before b_2(f+PH)-1 at the latest (f = first phase of step s); the first read of slot s+1
  //        comes after that barrier.
  //   WAR: stage(s+D) overwrites the slot of step s-1 in the first R of step s, i.e. after b_2f-1;
  //        the last reads of that slot were retired (lgkmcnt(0)) before b_2f-1 by both groups.
  const int total = n_my * nk;
  const int grp = wave >> 2;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < total) stage(d);
  {
    int ahead = total - 1;
    if (ahead > D - 1) ahead = D - 1;
    wait_stage(ahead, false);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if (grp == 1 && !ABL(8)) __builtin_amdgcn_s_barrier();
  // Waves 4-7 are the younger half of the workgroup and lose every arbitration (issue, LDS, address queue) to their SIMD
  // partners: their fragment reads and DMA issue took 1.9x / 1.25x the cycles of waves 0-3 (cycle stamps), and the older
  // half waited for them at the phase barriers.  One static priority raise for that half (cdna_hip_programming.md T5,
  // static form): fc1+GELU 1180 -> 1134 us, QKV 848 -> 833.  Only for the bf16-output epilogues: on the full-row tile it measured
  // +0.6 %, on the staged fp32 residual epilogue (ByteNet's residual GEMM) +1.3 %.
  if (grp == 1 && !ROW && !RES && !F32OUT) __builtin_amdgcn_s_setprio(SGPO_G1_PRIO);
  STAMP_DECL
  int slot = 0;   // ring slot of step s
  int s = 0;      // position in the flattened (tile, k-step) stream
  bool prev_full = false;
  // (tile loop around a k loop rather than one flat loop with a conditional epilogue: the register
  //  allocator then keeps the accumulators in place instead of shuffling them every step)
  // Epilogue placement: group 0 runs a tile's epilogue at the top of the next tile's first R phase;
  // group 1 (EARLY1) runs it at the end of the tile's last M phase, which is the SAME barrier interval,
  // so the two waves of a SIMD share the VALU for their epilogues instead of running them one after
  // the other.  Both orders put the stores between stage(e+D) and stage(e+1+D) in the wave's VM queue
  // (e = last step of the tile), so the wait counts below are the same for both groups.
  constexpr bool EARLY1 = !PRE && (!F32OUT || F32ST);
  for (int ti = 0; ti < n_my; ++ti) {
    STAMP(7);
    if (ti > 0 && !(EARLY1 && grp == 1)) {
      int pm0, pn0;
      tile_origin(ti - 1, pm0, pn0);
      prev_full = exact && tile_is_full(pm0, pn0) && !ABL(4);
      if (!ABL(4)) {
        int fslot = slot + D; if (fslot >= NST) fslot -= NST;   // the slot this step's prefetch fills next
        if (prev_full) epilogue(pm0, pn0, fslot, std::true_type{});   // stores older than the prefetch below
        else epilogue(pm0, pn0, fslot, std::false_type{});
      }
    }
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
   STAMP(5);
   auto step = [&](int kt, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;   // the tile's last k-step (peeled out of the loop)
    STAMP(7);
    if (s + D < total && !ABL(1)) {
      int ps = slot + D; if (ps >= NST) ps -= NST;
      stage(ps);
    }
    STAMP(0);
    if (PRE && LAST) {   // residual tile into registers, consumed by the epilogue one step later
      int m0, n0;
      tile_origin(ti, m0, n0);
      load_residual(m0, n0);
    }
    const char* base = smem + slot * STAGE;
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
      bf16x8 xf[MT][2], wf[NT][2];   // [32-row block][16-row half]
      {
        const int cxk = cx[ph];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int h = 0; h < 2; ++h) wf[t][h] = *reinterpret_cast<const bf16x8*>(base + b_lane + cxk + (t * 32 + h * 16) * ROWB);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int h = 0; h < 2; ++h) xf[t][h] = *reinterpret_cast<const bf16x8*>(base + a_lane + cxk + (t * 32 + h * 16) * ROWB);
      }
      if (ph == PH - 1 && s + 1 < total) {
        int ahead = total - 2 - s;
        if (ahead > D - 1) ahead = D - 1;
        // younger than stage s+1 besides the stages: the epilogue stores issued at kt = 0 (while
        // kt <= D-2), or the residual prefetch issued just above
        const bool ex_st = prev_full && ti > 0 && kt <= D - 2;
        const bool ex_ld = PRE && exact && LAST;
        wait_stage(ahead, ex_st || ex_ld);
      }
      STAMP_TAKE(st_a);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      STAMP_PUT(1, st_a);
      STAMP(2);
      if (!ABL(8)) __builtin_amdgcn_s_barrier();
      STAMP(3);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(2)) {
#pragma unroll
        for (int t = 0; t < NT; ++t) { asm volatile("" ::"v"(wf[t][0])); asm volatile("" ::"v"(wf[t][1])); }
#pragma unroll
        for (int t = 0; t < MT; ++t) { asm volatile("" ::"v"(xf[t][0])); asm volatile("" ::"v"(xf[t][1])); }
      } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
              for (int mh = 0; mh < 2; ++mh)
                acc[nt][mt][2 * nh + mh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][nh], xf[mt][mh], acc[nt][mt][2 * nh + mh], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      STAMP(4);
      if (EARLY1 && LAST && grp == 1 && ph == PH - 1) {
        int tm0, tn0;
        tile_origin(ti, tm0, tn0);
        prev_full = exact && tile_is_full(tm0, tn0) && !ABL(4);
        if (!ABL(4)) {
          // free slot = this step's own: every read of it was retired before this M phase began
          if (prev_full) epilogue(tm0, tn0, slot, std::true_type{});
          else epilogue(tm0, tn0, slot, std::false_type{});
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(5);
      }
      if (!ABL(8)) __builtin_amdgcn_s_barrier();
      STAMP(6);
      asm volatile("" ::: "memory");
    }
    if (++slot == NST) slot = 0;
    ++s;
   };
   for (int kt = 0; kt < nk - 1; ++kt) step(kt, std::false_type{});
   step(nk - 1, std::true_type{});
  }
  if constexpr (ROWSTAT) {
    // the epilogue holds barriers of its own: group 0 runs the last one where every other one runs, right after the
    // tile's last barrier pair and BEFORE its balancing barrier, so that its barriers pair with group 1's epilogue
    // (which ran at the end of group 1's last M phase); same free slot as group 1 used (slot(e) = slot + D)
    if (grp == 0) {
      int pm0, pn0, fslot = slot + D;
      if (fslot >= NST) fslot -= NST;
      tile_origin(n_my - 1, pm0, pn0);
      epilogue(pm0, pn0, fslot, std::false_type{});
      __builtin_amdgcn_s_barrier();
    }
    return;
  }
  if (grp == 0 && !ABL(8)) __builtin_amdgcn_s_barrier();
  if (!ABL(4) && !(EARLY1 && grp == 1)) {
    int pm0, pn0;
    tile_origin(n_my - 1, pm0, pn0);
    epilogue(pm0, pn0, 0, std::false_type{});   // the stream is over: every slot is free
  }
  STAMP(5);
  STAMP_FLUSH;
}

#ifdef SGPO_GEMM_STAMP
extern "C" int sgpo_gemm_stamps_read(unsigned long long* out, int n) {
  SGPO_CHECK_HIP(hipDeviceSynchronize());
  SGPO_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), (size_t)n * sizeof(unsigned long long)));
  return SGPO_OK;
}
#endif

template <int EPI, typename CFG, bool ROPE_LDS = false>
static int launch_cfg(const GemmParams& q, const GemmSeg& sg, const bf16_t* zero16, int n_cu, int dbg, hipStream_t stream) {
  constexpr int BM = CFG::WM * CFG::MT * 32, BN = CFG::WN * CFG::NT * 32;
  constexpr int RING = CFG::NST * (BM + BN) * CFG::BK * 2;
  constexpr bool F32OUT = (EPI == EPI_BIAS_F32 || EPI == EPI_ACC_F32);
  // segmented (dilated-convolution) operands exist for the epilogues ByteNet ends a convolution with
  constexpr bool CONV_OK = (EPI == EPI_BF16 || EPI == EPI_BIAS_F32 || EPI == EPI_LN_GELU);
  if (q.K % CFG::BK != 0 || sg.seg_k % CFG::BK != 0) return SGPO_ERR_UNSUPPORTED;
  int flags = PF_EXACT;
  size_t aux = 0;
  if (EPI == EPI_QKV_ROPE) {
    aux = (size_t)q.L * q.head_dim * sizeof(float);   // cos + sin
    if (!ROPE_LDS) { aux = 0; flags = 0; }           // table read from global: ordinary loads, drained waits
  } else if (!F32OUT || CFG::MT * CFG::NT * 4 > 16) {
    aux = (size_t)q.N * sizeof(float);
  }
  if (is_row_cfg<CFG>::value) aux = 0;   // the ring fills the LDS
  if (RING + aux > LDS_MAX_BYTES) return SGPO_ERR_UNSUPPORTED;
  const int ntm = (q.M + BM - 1) / BM, ntn = (q.N + BN - 1) / BN;
  const int tiles = ntm * ntn;
  const int grouped = (tiles > n_cu && n_cu % 8 == 0 && ntm >= 8) ? 1 : 0;
  const int grid = tiles > n_cu ? n_cu : tiles;
  const size_t lds = RING + ((aux + 15) & ~(size_t)15);
  static bool attr_set = false;
  if (!attr_set) {
    SGPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_p_kernel<EPI, false, CFG, ROPE_LDS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX_BYTES));
    if constexpr (CONV_OK)
      SGPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_p_kernel<EPI, true, CFG, ROPE_LDS>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX_BYTES));
    attr_set = true;
  }
  if (sg.n_seg > 1) {
    if constexpr (!CONV_OK) return SGPO_ERR_UNSUPPORTED;   // the caller falls back to the generic kernel (gemm.hip)
    else hipLaunchKernelGGL((gemm_p_kernel<EPI, true, CFG, ROPE_LDS>), dim3(grid), dim3(512), lds, stream, q, sg, zero16, ntm, ntn, grouped, flags, dbg);
  } else
    hipLaunchKernelGGL((gemm_p_kernel<EPI, false, CFG, ROPE_LDS>), dim3(grid), dim3(512), lds, stream, q, sg, zero16, ntm, ntn, grouped, flags, dbg);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

int launch_gemm_persistent(const GemmParams& p, int epi, const GemmSeg& sg, const bf16_t* zero16, hipStream_t stream) {
  // shapes the vectorised epilogues need; anything else falls back to the generic kernel
  const bool f32out = (epi == EPI_BIAS_F32 || epi == EPI_ACC_F32);
  const bool res = (epi == EPI_GATE_RES || epi == EPI_RES);
  if (!f32out && (p.N % 64 != 0 || p.ldc % 4 != 0)) return SGPO_ERR_UNSUPPORTED;
  if (epi == EPI_QKV_ROPE && (p.head_dim != 64 || p.rope_cols % 64 != 0)) return SGPO_ERR_UNSUPPORTED;
  if (res && (p.xin == nullptr || (p.gate == nullptr && epi == EPI_GATE_RES))) return SGPO_ERR_UNSUPPORTED;
  // few tiles: the 64/128-row kernel of gemm.hip fills more CUs (one guided batch of 10 sequences is 1190 rows)
  static int pmin = -1;
  if (pmin < 0) { const char* e = getenv("SGPO_GEMM_PMIN"); pmin = e ? atoi(e) : gemm_cu_count() / 4; }   // A/B: fewest 256x128 tiles for the persistent kernel
  if (p.M < 1024 || ((p.M + 255) / 256) * ((p.N + 127) / 128) < pmin) return SGPO_ERR_UNSUPPORTED;
  if (p.N > 4096) return SGPO_ERR_UNSUPPORTED;   // bias table in LDS beside the ring; zero page (16 KiB) for a null bias
  if (p.K % 64 != 0 || sg.seg_k % 64 != 0) return SGPO_ERR_UNSUPPORTED;
  static int n_cu = 0, tile_mode = -1, dbg = 0;
  if (!n_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t pr;
    n_cu = (hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
  }
  if (tile_mode < 0) {
    const char* e = getenv("SGPO_GEMM_TILE");   // "s": always the 256x128 tile (A/B measurements)
    tile_mode = (e && e[0] == 's') ? 1 : (e && e[0] == 'r') ? 2 : (e && e[0] == 'h') ? 3 : 0;
  }
#ifdef SGPO_GEMM_ABLATE
  { const char* e = getenv("SGPO_GEMM_DBG"); dbg = e ? atoi(e) : 0; }
  { const char* e = getenv("SGPO_GEMM_STAGGER_US"); dbg = (dbg & 255) | ((e ? atoi(e) : 0) << 8); }
#endif
#ifdef SGPO_GEMM_STAMP
  { const char* e = getenv("SGPO_GEMM_STAMP_EPI"); dbg = (e && atoi(e) == epi) ? 8 : 0; }
#endif
  GemmParams q = p;
  if (!q.bias) q.bias = reinterpret_cast<const float*>(zero16);
  // 256x256 or 256x128 tiles: a persistent launch takes ceil(tiles / CUs) rounds of one tile each, and a 256x128 tile
  // costs 0.625 of a 256x256 one (half the flops at 1.25x the operand bytes per flop).  With many tiles the big tile
  // wins (1.0 vs 1.25 per unit of work); at a few hundred sequences the round count decides (measured at B = 64 ... 800,
  // tools/dbg/gemm_bench.py with SGPO_GEMM_TILE: the rule picks the faster tile in every case).  Same bits either way.
  bool big = tile_mode == 0 && p.N % 256 == 0 && p.ldc % 4 == 0;
  if (big) {
    const long tm = (p.M + 255) / 256;
    const long r_big = (tm * ((p.N + 255) / 256) + n_cu - 1) / n_cu, r_res = (tm * ((p.N + 127) / 128) + n_cu - 1) / n_cu;
    big = 8 * r_big <= 5 * r_res;
  }
  if (epi == EPI_LN_GELU && (p.N != 512 || !p.ln_w || !p.ln_b || ((p.M + 127) / 128) < n_cu / 2)) return SGPO_ERR_UNSUPPORTED;
  if (epi == EPI_GATE_RES_LN) {
    if (p.N != 512 || p.ldc != 512 || !p.ln_w || !p.ln_shift || !p.ln_scale || !p.ln_out || !p.xin || !p.gate) return SGPO_ERR_UNSUPPORTED;
    // worth it only when the 128-row tiles fill their rounds: measured against GEMM + separate pass at B = 137 ... 1600
    // (L = 119), the fused kernel loses 4-15 % where the last round is mostly idle (fill 0.5-0.73) and wins from 0.79 up
    const long tiles = (p.M + 127) / 128, rounds = (tiles + n_cu - 1) / n_cu;
    if (tiles < 2 * n_cu || 4 * tiles < 3 * rounds * n_cu) return SGPO_ERR_UNSUPPORTED;
  }
  ProfScope prof_scope(epi == EPI_LN_GELU ? PK_GEMM_LN_GELU : epi == EPI_GATE_RES_LN ? PK_GEMM_GATE_RES_LN : PK_GEMM_BF16 + epi, stream,
                       2.0 * p.M * p.N * p.K);
  int rc = SGPO_ERR_UNSUPPORTED;
#ifdef SGPO_GEMM_ABLATE
  if (tile_mode == 2 && epi == EPI_GATE_RES && p.N == 512)   // K loop of the full-row tile, no epilogue (bit 4 forced)
    return launch_cfg<EPI_GATE_RES, CfgRow>(q, sg, zero16, n_cu, dbg | 4, stream);
  if (tile_mode == 2 && epi == EPI_BIAS_GELU && p.N % 512 == 0)   // the same on the fc1 shape: the fc1 phase of a fused MLP (round 3)
    return launch_cfg<EPI_BIAS_GELU, CfgRow>(q, sg, zero16, n_cu, dbg | 4, stream);
  if (tile_mode == 3 && (epi == EPI_GATE_RES || epi == EPI_BIAS_GELU) && p.N % 512 == 0) {   // K loop of the 64 x 512 tile
    if (epi == EPI_GATE_RES) return launch_cfg<EPI_GATE_RES, CfgHalf>(q, sg, zero16, n_cu, dbg | 4, stream);
    return launch_cfg<EPI_BIAS_GELU, CfgHalf>(q, sg, zero16, n_cu, dbg | 4, stream);
  }
#endif
  if (epi == EPI_LN_GELU) return launch_cfg<EPI_LN_GELU, CfgRow>(q, sg, zero16, n_cu, dbg, stream);
  if (epi == EPI_GATE_RES_LN) return launch_cfg<EPI_GATE_RES_LN, CfgRow>(q, sg, zero16, n_cu, dbg, stream);
#define LAUNCH_BOTH(E)                                                              \
  case E:                                                                           \
    if (big) rc = launch_cfg<E, CfgBig>(q, sg, zero16, n_cu, dbg, stream);          \
    if (rc == SGPO_ERR_UNSUPPORTED) rc = launch_cfg<E, CfgRes>(q, sg, zero16, n_cu, dbg, stream); \
    break;
#define LAUNCH_RES(E) \
  case E: rc = launch_cfg<E, CfgRes>(q, sg, zero16, n_cu, dbg, stream); break;
  switch (epi) {
    LAUNCH_BOTH(EPI_BF16)
    case EPI_QKV_ROPE:
      if (big) {
        rc = launch_cfg<EPI_QKV_ROPE, CfgBig, true>(q, sg, zero16, n_cu, dbg, stream);
        if (rc == SGPO_ERR_UNSUPPORTED) rc = launch_cfg<EPI_QKV_ROPE, CfgBig, false>(q, sg, zero16, n_cu, dbg, stream);
      }
      if (rc == SGPO_ERR_UNSUPPORTED) rc = launch_cfg<EPI_QKV_ROPE, CfgRes, false>(q, sg, zero16, n_cu, dbg, stream);
      break;
    LAUNCH_BOTH(EPI_BIAS_GELU)
    LAUNCH_RES(EPI_BIAS_GELU_ERF)   // erff needs too many registers beside the 256x256 accumulators
    LAUNCH_BOTH(EPI_GATE_RES)
    LAUNCH_BOTH(EPI_RES)
    LAUNCH_BOTH(EPI_BIAS_F32)
    LAUNCH_RES(EPI_ACC_F32)    // the staged form spills beside the 256x256 accumulators; unused on the hot paths
    default: break;
  }
#undef LAUNCH_BOTH
#undef LAUNCH_RES
  return rc;
}

int launch_gemm_ln_gelu(const GemmParams& p, int n_seg, int seg_k, const int* shifts, hipStream_t stream) {
  if (p.M <= 0 || p.K <= 0 || p.K % 64 != 0 || n_seg < 1 || n_seg > 8 || seg_k % 64 != 0 || n_seg * seg_k != p.K ||
      p.lda % 8 != 0 || p.ldw % 8 != 0 || p.ldc % 8 != 0 || p.L <= 0)
    return SGPO_ERR_UNSUPPORTED;
  GemmSeg sg;
  sg.n_seg = n_seg; sg.seg_k = seg_k;
  for (int i = 0; i < 8; ++i) sg.shift[i] = (shifts && i < n_seg) ? shifts[i] : 0;
  const bf16_t* z = nullptr;
  const int rc = gemm_get_zero16(&z);
  if (rc) return rc;
  return launch_gemm_persistent(p, EPI_LN_GELU, sg, z, stream);
}

int launch_gemm_gate_res_ln(const GemmParams& p, hipStream_t stream) {
  if (p.M <= 0 || p.K <= 0 || p.K % 64 != 0 || p.lda % 8 != 0 || p.ldw % 8 != 0 || p.L <= 0) return SGPO_ERR_UNSUPPORTED;
  GemmSeg sg;
  sg.n_seg = 1; sg.seg_k = p.K;
  for (int i = 0; i < 8; ++i) sg.shift[i] = 0;
  const bf16_t* z = nullptr;
  const int rc = gemm_get_zero16(&z);
  if (rc) return rc;
  return launch_gemm_persistent(p, EPI_GATE_RES_LN, sg, z, stream);
}
