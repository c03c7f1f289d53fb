// Optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
// Disabled by default: the macros cost one branch.  Events come from a preallocated pool
// (sgpo_prof_enable allocates it, outside any hot path); when the pool is exhausted further
// launches are simply not timed.
#pragma once
#include <hip/hip_runtime.h>

enum ProfKind {
  PK_GEMM_BF16 = 0, PK_GEMM_QKV_ROPE, PK_GEMM_GATE_RES, PK_GEMM_BIAS_GELU, PK_GEMM_BIAS_F32, PK_GEMM_GELU_ERF,
  PK_GEMM_RES, PK_GEMM_ACC, PK_ATTENTION, PK_LN_MOD, PK_EMBED, PK_REVERSE_STEP, PK_MLP_BASE, PK_MLP_JUMP,
  PK_D3PM_POSTERIOR, PK_DAPS_MH, PK_MISC, PK_FITNESS, PK_GEMM_LN_GELU, PK_GEMM_GATE_RES_LN, PK_COUNT
};

bool prof_on();
void prof_begin(int kind, hipStream_t st, double work);
void prof_end(hipStream_t st);

struct ProfScope {
  hipStream_t st; bool on;
  ProfScope(int kind, hipStream_t s, double work) : st(s), on(prof_on()) { if (on) prof_begin(kind, s, work); }
  ~ProfScope() { if (on) prof_end(st); }
};
