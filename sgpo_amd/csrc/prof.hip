#include "prof.h"

#include <vector>

#include "../../include/sgpo_hip.h"
#include "common.h"

namespace {
struct Rec { int kind; double work; hipEvent_t a, b; };
bool g_on = false;
std::vector<hipEvent_t> g_pool;
size_t g_next = 0;
std::vector<Rec> g_recs;
Rec g_cur;
bool g_open = false;
}  // namespace

bool prof_on() { return g_on; }

void prof_begin(int kind, hipStream_t st, double work) {
  if (g_next + 2 > g_pool.size()) { g_open = false; return; }
  g_cur.kind = kind; g_cur.work = work;
  g_cur.a = g_pool[g_next++]; g_cur.b = g_pool[g_next++];
  (void)hipEventRecord(g_cur.a, st);
  g_open = true;
}
void prof_end(hipStream_t st) {
  if (!g_open) return;
  (void)hipEventRecord(g_cur.b, st);
  g_recs.push_back(g_cur);
  g_open = false;
}

extern "C" int sgpo_prof_enable(int on, int max_launches) {
  if (on) {
    size_t want = (size_t)(max_launches > 0 ? max_launches : 0) * 2;
    while (g_pool.size() < want) {
      hipEvent_t e;
      SGPO_CHECK_HIP(hipEventCreate(&e));
      g_pool.push_back(e);
    }
    g_next = 0; g_recs.clear();
    g_recs.reserve(want / 2);   // no host allocation while recording either
  }
  g_on = on != 0;
  return SGPO_OK;
}

extern "C" int sgpo_prof_reset(void) { g_next = 0; g_recs.clear(); return SGPO_OK; }

// Synchronises the device, then sums per kind: total ms, launches, total algorithmic work.
extern "C" int sgpo_prof_read(int n_kinds, double* total_ms, int64_t* launches, double* work) {
  SGPO_REQUIRE(total_ms && launches && work && n_kinds >= PK_COUNT, "prof_read: need %d slots", (int)PK_COUNT);
  SGPO_CHECK_HIP(hipDeviceSynchronize());
  for (int i = 0; i < n_kinds; ++i) { total_ms[i] = 0; launches[i] = 0; work[i] = 0; }
  for (const Rec& r : g_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    total_ms[r.kind] += ms; launches[r.kind] += 1; work[r.kind] += r.work;
  }
  return SGPO_OK;
}

extern "C" const char* sgpo_prof_kind_name(int kind) {
  static const char* names[PK_COUNT] = {
      "gemm_bf16_kernel<EPI_BF16>", "gemm_bf16_kernel<EPI_QKV_ROPE>", "gemm_bf16_kernel<EPI_GATE_RES>",
      "gemm_bf16_kernel<EPI_BIAS_GELU>", "gemm_bf16_kernel<EPI_BIAS_F32>", "gemm_bf16_kernel<EPI_BIAS_GELU_ERF>",
      "gemm_bf16_kernel<EPI_RES>", "gemm_bf16_kernel<EPI_ACC_F32>", "attention_kernel", "ln_mod_kernel",
      "embed_kernel", "reverse_step_kernel", "mlp_base_kernel", "mlp_jump_kernel", "d3pm_posterior_kernel",
      "daps_mh_kernel", "misc", "fitness_member_kernel", "gemm_bf16_kernel<EPI_LN_GELU>", "gemm_bf16_kernel<EPI_GATE_RES_LN>"};
  return (kind >= 0 && kind < PK_COUNT) ? names[kind] : "?";
}
