// Shared device helpers for the SGPO sampling-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;  // storage type in HBM

#define SGPO_OK 0
#define SGPO_ERR_ARG (-1)
#define SGPO_ERR_HIP (-2)
#define SGPO_ERR_WORKSPACE (-3)
#define SGPO_ERR_UNSUPPORTED (-4)

extern "C" void sgpo_set_error(const char* fmt, ...);

#define SGPO_CHECK_HIP(expr)                                                      \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      sgpo_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return SGPO_ERR_HIP;                                                        \
    }                                                                             \
  } while (0)

#define SGPO_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      sgpo_set_error(__VA_ARGS__);     \
      return SGPO_ERR_ARG;             \
    }                                  \
  } while (0)

// round-to-nearest-even f32 -> bf16 (NaN stays NaN: plain cast lowers to v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}
// two f32 -> one dword of two bf16 (lo = a, hi = b) in ONE v_cvt_pk_bf16_f32; the same round-to-nearest-even values as
// two f2bf (written as `f2bf(a) | f2bf(b) << 16` the compiler emits two conversions, a shift and an or)
typedef __bf16 bf16x2_vt __attribute__((ext_vector_type(2)));
typedef float f32x2_vt __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  const bf16x2_vt v = __builtin_convertvector(f32x2_vt{a, b}, bf16x2_vt);
  return *reinterpret_cast<const uint32_t*>(&v);
}
__device__ __forceinline__ float bf2f(bf16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// reductions inside each 32-lane half of the wave
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float half_max(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---------------------------------------------------------------------------
// Philox4x32-10, counter-based.  key = (seed lo, seed hi); counter =
// (element index lo, element index hi, call offset lo, call offset hi).
// Shard invariance: the element index is the GLOBAL (row, k) index.
struct Philox4 { uint32_t x, y, z, w; };
__device__ __forceinline__ Philox4 philox4x32_10(uint64_t seed, uint64_t idx, uint64_t offset) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32);
  uint32_t c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}
// uniform in (0,1): 24 random bits, centred
__device__ __forceinline__ float u01_open(uint32_t w) {
  return ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f);
}
