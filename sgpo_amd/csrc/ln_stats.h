// LayerNorm row statistics over H = 512 in ONE fixed summation order, shared by the separate pass (ln_mod_kernel<512>,
// dit_kernels.hip) and by the fused GEMM epilogue (gemm_p.hip, EPI_GATE_RES_LN), so that both give the same bits: a batch
// small enough for the 64/128-row GEMM + separate pass must reproduce the rows of a big batch through the fused kernel.
//
// The row is cut into eight 64-column segments (= the eight waves of the full-row GEMM tile).  Inside a segment a lane
// holds one quad of 4 consecutive columns, 16 lanes of one DPP row hold the segment:
//   quad sum   (x + y) + (z + w)
//   segment    DPP tree over the 16 lanes (lane^1, lane^2, mirror in 8, mirror in 16): S, mean_s = S / 64
//   M2_s       sum of squared deviations from mean_s: per quad fma chain x, y, z, w from 0, then the same tree
// and the eight (mean_s, M2_s) are combined in segment order (Chan et al.: exact, as accurate as two passes):
//   mean = (sum_s mean_s) / 8,   M2 = sum_s M2_s + 64 * sum_s (mean_s - mean)^2,   rstd = rsqrt(M2 / 512 + eps).
#pragma once
#include "common.h"
#include "gemm.h"

#define LNS_DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ float lns_row16_sum(float v) {   // every lane of the 16-lane row gets the same total
  v = rn_add(v, LNS_DPP_F(v, 0xB1));
  v = rn_add(v, LNS_DPP_F(v, 0x4E));
  v = rn_add(v, LNS_DPP_F(v, 0x141));
  v = rn_add(v, LNS_DPP_F(v, 0x140));
  return v;
}
__device__ __forceinline__ float lns_quad_sum(float x, float y, float z, float w) { return rn_add(rn_add(x, y), rn_add(z, w)); }
__device__ __forceinline__ float lns_quad_sq(float x, float y, float z, float w, float mean) {
  const float dx = x - mean, dy = y - mean, dz = z - mean, dw = w - mean;
  return fmaf(dw, dw, fmaf(dz, dz, fmaf(dy, dy, rn_mul(dx, dx))));
}
// segment statistics of the 64 columns held by a 16-lane row (one quad per lane)
__device__ __forceinline__ void lns_segment(float x, float y, float z, float w, float& mean_s, float& m2_s) {
  mean_s = rn_mul(lns_row16_sum(lns_quad_sum(x, y, z, w)), 1.0f / 64.0f);
  m2_s = lns_row16_sum(lns_quad_sq(x, y, z, w, mean_s));
}
__device__ __forceinline__ void lns_combine8(const float (&ms)[8], const float (&qs)[8], float eps, float& mean, float& rstd) {
  float sm = 0.f, m2 = 0.f;
#pragma unroll
  for (int s = 0; s < 8; ++s) { sm = rn_add(sm, ms[s]); m2 = rn_add(m2, qs[s]); }
  mean = rn_mul(sm, 0.125f);
  float dv = 0.f;
#pragma unroll
  for (int s = 0; s < 8; ++s) { const float d = ms[s] - mean; dv = fmaf(d, d, dv); }
  m2 = fmaf(64.0f, dv, m2);
  rstd = rsqrtf(fmaf(m2, 1.0f / 512.0f, eps));
}
// adaLN modulate of one normalised element: ((v - mean) rstd w) (1 + scale) + shift
__device__ __forceinline__ float lns_modulate(float v, float mean, float rstd, float w, float scale, float shift) {
  return fmaf(rn_mul(rn_mul(v - mean, rstd), w), 1.0f + scale, shift);
}
