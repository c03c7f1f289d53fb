// Row f-2: what the reference does to a sampled batch between the sampler and the fitness oracle, on
// token tensors instead of Python strings.
//   * project_sequences  (sampling/base.py:40-63): special tokens -> a random standard residue, one draw
//     per distinct special token of a sequence (str.replace replaces every occurrence); optionally at most
//     n_max_mutations of the positions that differ from the parent are kept, chosen uniformly at random.
//   * ProteinPredictorDataset.update_data (dataset/protein.py:153-213): de-duplication inside the batch and
//     against the history (BO) or against the current round, truncation to n_needed, append to the store.
//   * the 4-site landscape lookup of update_data (dataset/protein.py:188-197).
// The set is an open-addressing table in HBM: one 64-bit word per slot = 40 hash bits | 24-bit index of a
// representative sequence; a hash match is always confirmed by comparing the tokens, so the result is exact.
// HBM-bound integer work: one wave per sequence, tokens read once, coalesced.
#include "common.h"
#include <new>

namespace {

constexpr uint64_t ACQ_EMPTY = ~0ull;
constexpr uint64_t ACQ_TOMB = ~0ull - 1;
constexpr uint32_t ACQ_REP_MASK = 0xFFFFFFu;
constexpr uint32_t ACQ_NONE = 0xFFFFFFFFu;

__device__ __forceinline__ uint64_t mix64(uint64_t z) {   // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)v, o), hi = __shfl_xor((uint32_t)(v >> 32), o);
    v += ((uint64_t)hi << 32) | lo;
  }
  return v;
}
__device__ __forceinline__ uint64_t bcast_u64(uint64_t v) {
  const uint32_t lo = __shfl((uint32_t)v, 0), hi = __shfl((uint32_t)(v >> 32), 0);
  return ((uint64_t)hi << 32) | lo;
}

// ---- projection ------------------------------------------------------------------------------------
// one wave per sequence; L <= 1024 (keys and flags of the sequence live in LDS)
constexpr int PROJ_MAX_L = 1024;
__global__ __launch_bounds__(256) void project_kernel(int64_t* __restrict__ tokens, const int64_t* __restrict__ full_seq,
                                                      int N, int L, int n_std, int n_special,
                                                      const float* __restrict__ u_special, int n_max,
                                                      const float* __restrict__ keys, int* __restrict__ bad_flag) {
  __shared__ float s_key[4][PROJ_MAX_L];
  __shared__ unsigned char s_mut[4][PROJ_MAX_L];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;   // wave-uniform; no block-wide barrier below
  int64_t* row = tokens + (size_t)n * L;
  int n_mut = 0;
  for (int j = lane; j < L; j += 64) {
    int64_t tok = row[j];
    if (tok < 0 || tok >= n_std + n_special) {
      if (bad_flag) atomicOr(bad_flag, 1);
      tok = 0;
    }
    if (tok >= n_std) {
      if (u_special) {
        int r = (int)(u_special[(size_t)n * n_special + (tok - n_std)] * (float)n_std);
        tok = r < n_std - 1 ? r : n_std - 1;
      } else if (bad_flag) {
        atomicOr(bad_flag, 2);   // a special token and no noise to replace it with
      }
      row[j] = tok;
    }
    if (n_max >= 0) {
      const bool m = tok != full_seq[j];
      s_mut[wave][j] = m;
      s_key[wave][j] = keys[(size_t)n * L + j];
      n_mut += m;
    }
  }
  if (n_max < 0) return;
  n_mut = (int)wave_sum((float)n_mut);   // <= 1024: exact in fp32
  if (n_mut <= n_max) return;
  __builtin_amdgcn_wave_barrier();       // LDS operations of one wave execute in order: its writes are visible below
  for (int j = lane; j < L; j += 64) {
    if (!s_mut[wave][j]) continue;
    const float kj = s_key[wave][j];
    int rank = 0;
    for (int k = 0; k < L; ++k) {
      const float kk = s_key[wave][k];
      rank += s_mut[wave][k] && (kk < kj || (kk == kj && k < j));
    }
    if (rank >= n_max) row[j] = full_seq[j];
  }
}

// ---- acquisition set ---------------------------------------------------------------------------------
struct AcqDev {
  uint64_t* slots;       // [table]  40 hash bits << 24 | representative (store index, or n_store + batch row)
  uint32_t* first;       // [table]  smallest batch row that hit the slot during the running update
  int* slot_round;       // [table]  last round in which the slot's sequence was appended
  unsigned char* store;  // [capacity][L]
  int* store_round;      // [capacity]
  uint32_t mask;
  int L;
};

__device__ __forceinline__ uint64_t seq_hash(const int64_t* row, int L, int lane) {
  uint64_t h = 0;
  for (int j = lane; j < L; j += 64) h += mix64((uint64_t)row[j] + ((uint64_t)(j + 1) << 32));
  h = mix64(wave_sum_u64(h));
  return h;
}

// one wave per batch row: find or claim the row's slot
__global__ __launch_bounds__(256) void acq_insert_kernel(AcqDev d, const int64_t* __restrict__ tokens, int N, uint32_t n_store,
                                                         uint32_t* __restrict__ slot_of, int* __restrict__ err) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= N) return;
  const int64_t* row = tokens + (size_t)i * d.L;
  const uint64_t h = seq_hash(row, d.L, lane);
  const uint64_t hi40 = h >> 24;
  const uint64_t mine = (hi40 << 24) | (uint64_t)(n_store + (uint32_t)i);
  uint32_t slot = (uint32_t)(h ^ (h >> 40)) & d.mask;
  uint32_t probes = 0;
  bool found = false;
  while (!found) {
    uint64_t cur = 0;
    if (lane == 0) {
      cur = __hip_atomic_load(&d.slots[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == ACQ_EMPTY) {
        uint64_t expect = ACQ_EMPTY;
        if (__hip_atomic_compare_exchange_strong(&d.slots[slot], &expect, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT))
          cur = mine;
        else
          cur = expect;
      }
    }
    cur = bcast_u64(cur);
    if (cur == mine) { found = true; break; }
    if (cur != ACQ_TOMB && (cur >> 24) == hi40) {
      const uint32_t rep = (uint32_t)cur & ACQ_REP_MASK;
      bool eq = true;
      if (rep < n_store) {
        const unsigned char* o = d.store + (size_t)rep * d.L;
        for (int j = lane; j < d.L; j += 64) eq &= (int64_t)o[j] == row[j];
      } else {
        const int64_t* o = tokens + (size_t)(rep - n_store) * d.L;
        for (int j = lane; j < d.L; j += 64) eq &= o[j] == row[j];
      }
      if (__all(eq)) { found = true; break; }
    }
    slot = (slot + 1) & d.mask;
    if (++probes > d.mask) {   // table full: sized wrongly by the host
      if (lane == 0) atomicOr(err, 1);
      break;
    }
  }
  if (lane == 0) {
    slot_of[i] = found ? slot : ACQ_NONE;
    if (found) atomicMin(&d.first[slot], (uint32_t)i);
  }
}

// scope: 0 keep every row, 1 drop rows already in the store (any round), 2 drop rows appended in `round`.
// flag bit 0 = keep (before truncation), bit 1 = first row of the batch on its slot
__global__ void acq_decide_kernel(AcqDev d, int N, uint32_t n_store, int scope, int round,
                                  const uint32_t* __restrict__ slot_of, unsigned char* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const uint32_t slot = slot_of[i];
  if (slot == ACQ_NONE) { flag[i] = 0; return; }
  const uint32_t rep = (uint32_t)d.slots[slot] & ACQ_REP_MASK;
  const bool existed = rep < n_store;
  const bool winner = d.first[slot] == (uint32_t)i;
  const bool in_scope = scope == 1 ? existed : scope == 2 ? (existed && d.slot_round[slot] == round) : false;
  const bool keep = scope == 0 ? true : (winner && !in_scope);
  flag[i] = (unsigned char)((keep ? 1 : 0) | (winner ? 2 : 0));
}

// stable compaction of the kept rows, truncated to n_needed: one workgroup of 1024 threads
__global__ __launch_bounds__(1024) void acq_scan_kernel(int N, int n_needed, const unsigned char* __restrict__ flag,
                                                        int* __restrict__ rank, int64_t* __restrict__ kept_idx,
                                                        int* __restrict__ counts) {
  __shared__ int s_part[1024];
  const int t = threadIdx.x;
  const int per = (N + 1023) / 1024;
  const int b = t * per, e = min(b + per, N);
  int c = 0;
  for (int i = b; i < e; ++i) c += flag[i] & 1;
  s_part[t] = c;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = t >= o ? s_part[t - o] : 0;
    __syncthreads();
    s_part[t] += v;
    __syncthreads();
  }
  int r = s_part[t] - c;
  for (int i = b; i < e; ++i) {
    if (flag[i] & 1) {
      rank[i] = r;
      if (r < n_needed) kept_idx[r] = i;
      ++r;
    } else {
      rank[i] = -1;
    }
  }
  if (t == 1023) {
    counts[0] = min(s_part[1023], n_needed);   // appended
    counts[1] = s_part[1023];                  // unique / admissible before truncation
  }
}

// append the kept rows to the store and settle the slots this batch touched; one wave per batch row
__global__ __launch_bounds__(256) void acq_commit_kernel(AcqDev d, const int64_t* __restrict__ tokens, int N, uint32_t n_store,
                                                         int n_needed, int round, const uint32_t* __restrict__ slot_of,
                                                         const unsigned char* __restrict__ flag, const int* __restrict__ rank,
                                                         int* __restrict__ n_tomb) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= N) return;
  const uint32_t slot = slot_of[i];
  if (slot == ACQ_NONE) return;
  const unsigned char f = flag[i];
  const int r = rank[i];
  const bool kept = (f & 1) && r < n_needed;
  if (kept) {
    const int64_t* row = tokens + (size_t)i * d.L;
    unsigned char* o = d.store + (size_t)(n_store + r) * d.L;
    for (int j = lane; j < d.L; j += 64) o[j] = (unsigned char)row[j];
    if (lane == 0) d.store_round[n_store + r] = round;
  }
  if (lane == 0 && (f & 2)) {   // the slot's first row of this batch settles it
    const uint64_t w = d.slots[slot];
    const uint32_t rep = (uint32_t)w & ACQ_REP_MASK;
    if (rep >= n_store) {       // claimed by this batch
      d.slots[slot] = kept ? ((w & ~(uint64_t)ACQ_REP_MASK) | (uint64_t)(n_store + r)) : ACQ_TOMB;
      if (kept) d.slot_round[slot] = round;
      else atomicAdd(n_tomb, 1);   // a claimed slot given up (row cut off by n_needed, or a batch that did not fit)
    } else if (kept) {
      d.slot_round[slot] = round;
    }
    d.first[slot] = ACQ_NONE;   // nobody reads `first` in this kernel
  }
}

// Tombstones are never claimed again by acq_insert_kernel (two rows with the same tokens could then settle on
// different slots), so they only lengthen probe chains.  When they pile up the table is wiped and refilled from the
// store: one wave per store row, same probing as acq_insert_kernel; rows with equal tokens (scope 0 keeps duplicates)
// share the slot of whichever claimed it first, slot_round = the latest round among them.
__global__ __launch_bounds__(256) void acq_rebuild_kernel(AcqDev d, uint32_t n_store, int* __restrict__ err) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t i = blockIdx.x * 4 + wave;
  if (i >= n_store) return;
  const unsigned char* row = d.store + (size_t)i * d.L;
  uint64_t h = 0;
  for (int j = lane; j < d.L; j += 64) h += mix64((uint64_t)row[j] + ((uint64_t)(j + 1) << 32));   // == seq_hash
  h = mix64(wave_sum_u64(h));
  const uint64_t hi40 = h >> 24;
  const uint64_t mine = (hi40 << 24) | (uint64_t)i;
  uint32_t slot = (uint32_t)(h ^ (h >> 40)) & d.mask;
  uint32_t probes = 0;
  bool found = false;
  while (!found) {
    uint64_t cur = 0;
    if (lane == 0) {
      cur = __hip_atomic_load(&d.slots[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == ACQ_EMPTY) {
        uint64_t expect = ACQ_EMPTY;
        cur = __hip_atomic_compare_exchange_strong(&d.slots[slot], &expect, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT) ? mine : expect;
      }
    }
    cur = bcast_u64(cur);
    if (cur == mine) { found = true; break; }
    if ((cur >> 24) == hi40) {
      const unsigned char* o = d.store + (size_t)((uint32_t)cur & ACQ_REP_MASK) * d.L;
      bool eq = true;
      for (int j = lane; j < d.L; j += 64) eq &= o[j] == row[j];
      if (__all(eq)) { found = true; break; }
    }
    slot = (slot + 1) & d.mask;
    if (++probes > d.mask) {
      if (lane == 0) atomicOr(err, 1);
      break;
    }
  }
  if (found && lane == 0) atomicMax(&d.slot_round[slot], d.store_round[i]);
}

__global__ void acq_read_kernel(const unsigned char* __restrict__ store, size_t n, int64_t* __restrict__ out) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = store[i];
}

// 4-site landscape: index = sum_k tok[residues[k]] * n_std^(n_res-1-k); missing combos hold `missing`
__global__ void landscape_kernel(const float* __restrict__ table, const int64_t* __restrict__ tokens, int N, int L,
                                 const int* __restrict__ residues, int n_res, int n_std, float* __restrict__ out,
                                 int* __restrict__ bad_flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  size_t idx = 0;
  bool bad = false;
  for (int k = 0; k < n_res; ++k) {
    const int pos = residues[k];
    if (pos < 0 || pos >= L) { bad = true; continue; }   // a position outside the sequence: flagged, never read
    const int64_t tok = tokens[(size_t)i * L + pos];
    bad |= tok < 0 || tok >= n_std;
    idx = idx * n_std + (size_t)(bad ? 0 : tok);
  }
  if (bad && bad_flag) atomicOr(bad_flag, 1);
  out[i] = bad ? 0.f : table[idx];
}

}  // namespace

struct sgpo_acq {
  AcqDev d;
  size_t capacity, table;
  uint32_t n_store;
  int max_batch;
  uint32_t* slot_of;
  unsigned char* flag;
  int* rank;
  int* counts;   // [0] appended, [1] admissible, [2] error bits, [3] tombstones in the table
};

extern "C" int sgpo_project_tokens(int64_t* tokens, const int64_t* full_seq, int N, int L, int n_std, int n_special,
                                   const float* u_special, int n_max_mutations, const float* keys, int* bad_flag,
                                   void* stream) {
  SGPO_REQUIRE(N >= 0 && L > 0 && L <= PROJ_MAX_L, "project_tokens: need 0 < L <= %d (got %d)", PROJ_MAX_L, L);
  SGPO_REQUIRE(n_std > 0 && n_special >= 0, "project_tokens: bad alphabet sizes");
  SGPO_REQUIRE(n_max_mutations < 0 || (keys && full_seq), "project_tokens: n_max_mutations needs keys and full_seq");
  if (N == 0) return SGPO_OK;
  hipLaunchKernelGGL(project_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, tokens, full_seq, N, L, n_std,
                     n_special, u_special, n_max_mutations < 0 ? -1 : n_max_mutations, keys, bad_flag);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

extern "C" void sgpo_acq_destroy(sgpo_acq* a);

extern "C" int sgpo_acq_create(int L, size_t capacity, int max_batch, sgpo_acq** out) {
  SGPO_REQUIRE(out && L > 0 && capacity > 0 && max_batch > 0, "acq_create: bad arguments");
  SGPO_REQUIRE(capacity + (size_t)max_batch < ACQ_REP_MASK, "acq_create: capacity + max_batch must stay below 2^24");
  sgpo_acq* a = new (std::nothrow) sgpo_acq();
  SGPO_REQUIRE(a, "acq_create: out of host memory");
  a->capacity = capacity;
  a->max_batch = max_batch;
  a->n_store = 0;
  size_t table = 1024;
  while (table < 2 * (capacity + (size_t)max_batch)) table <<= 1;
  a->table = table;
  a->d.L = L;
  a->d.mask = (uint32_t)(table - 1);
  // a failure part-way releases what was allocated (the handle is value-initialised: unset pointers are null)
  auto build = [&]() -> int {
    SGPO_CHECK_HIP(hipMalloc(&a->d.slots, table * 8));
    SGPO_CHECK_HIP(hipMalloc(&a->d.first, table * 4));
    SGPO_CHECK_HIP(hipMalloc(&a->d.slot_round, table * 4));
    SGPO_CHECK_HIP(hipMalloc(&a->d.store, capacity * (size_t)L));
    SGPO_CHECK_HIP(hipMalloc(&a->d.store_round, capacity * 4));
    SGPO_CHECK_HIP(hipMalloc(&a->slot_of, (size_t)max_batch * 4));
    SGPO_CHECK_HIP(hipMalloc(&a->flag, (size_t)max_batch));
    SGPO_CHECK_HIP(hipMalloc(&a->rank, (size_t)max_batch * 4));
    SGPO_CHECK_HIP(hipMalloc(&a->counts, 16));
    SGPO_CHECK_HIP(hipMemset(a->d.slots, 0xFF, table * 8));
    SGPO_CHECK_HIP(hipMemset(a->d.first, 0xFF, table * 4));
    SGPO_CHECK_HIP(hipMemset(a->d.slot_round, 0xFF, table * 4));
    SGPO_CHECK_HIP(hipMemset(a->counts, 0, 16));
    return SGPO_OK;
  };
  const int rc = build();
  if (rc != SGPO_OK) {
    sgpo_acq_destroy(a);
    return rc;
  }
  *out = a;
  return SGPO_OK;
}

extern "C" void sgpo_acq_destroy(sgpo_acq* a) {
  if (!a) return;
  void* bufs[] = {a->d.slots, a->d.first, a->d.slot_round, a->d.store, a->d.store_round, a->slot_of, a->flag, a->rank, a->counts};
  for (void* b : bufs) (void)hipFree(b);
  delete a;
}

extern "C" long long sgpo_acq_count(const sgpo_acq* a) { return a ? (long long)a->n_store : -1; }

static int acq_rebuild(sgpo_acq* a, hipStream_t s) {
  SGPO_CHECK_HIP(hipMemsetAsync(a->d.slots, 0xFF, a->table * 8, s));
  SGPO_CHECK_HIP(hipMemsetAsync(a->d.first, 0xFF, a->table * 4, s));
  SGPO_CHECK_HIP(hipMemsetAsync(a->d.slot_round, 0xFF, a->table * 4, s));
  SGPO_CHECK_HIP(hipMemsetAsync(a->counts + 3, 0, 4, s));
  if (a->n_store)
    hipLaunchKernelGGL(acq_rebuild_kernel, dim3((a->n_store + 3) / 4), dim3(256), 0, s, a->d, a->n_store, a->counts + 2);
  SGPO_CHECK_HIP(hipGetLastError());
  int err = 0;
  SGPO_CHECK_HIP(hipMemcpyAsync(&err, a->counts + 2, sizeof(int), hipMemcpyDeviceToHost, s));
  SGPO_CHECK_HIP(hipStreamSynchronize(s));
  if (err) {
    SGPO_CHECK_HIP(hipMemsetAsync(a->counts + 2, 0, 4, s));
    sgpo_set_error("acq_update: hash table full while sweeping tombstones (table %zu slots)", a->table);
    return SGPO_ERR_WORKSPACE;
  }
  return SGPO_OK;
}

extern "C" int sgpo_acq_update(sgpo_acq* a, const int64_t* tokens, int N, int n_needed, int scope, int round,
                               int64_t* kept_idx, int* n_kept, int* n_admissible, void* stream) {
  SGPO_REQUIRE(a && n_kept, "acq_update: null handle or n_kept");
  SGPO_REQUIRE(N >= 0 && N <= a->max_batch, "acq_update: batch of %d rows exceeds max_batch %d", N, a->max_batch);
  SGPO_REQUIRE(scope >= 0 && scope <= 2, "acq_update: scope must be 0, 1 or 2");
  SGPO_REQUIRE(N == 0 || (tokens && kept_idx), "acq_update: null tokens or kept_idx");
  if (n_needed < 0) n_needed = 0;
  *n_kept = 0;
  if (n_admissible) *n_admissible = 0;
  if (N == 0) return SGPO_OK;
  const size_t room = a->capacity - a->n_store;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(acq_insert_kernel, dim3((N + 3) / 4), dim3(256), 0, s, a->d, tokens, N, a->n_store, a->slot_of,
                     a->counts + 2);
  hipLaunchKernelGGL(acq_decide_kernel, dim3((N + 255) / 256), dim3(256), 0, s, a->d, N, a->n_store, scope, round,
                     a->slot_of, a->flag);
  hipLaunchKernelGGL(acq_scan_kernel, dim3(1), dim3(1024), 0, s, N, n_needed, a->flag, a->rank, kept_idx, a->counts);
  SGPO_CHECK_HIP(hipGetLastError());
  int host[3];
  SGPO_CHECK_HIP(hipMemcpyAsync(host, a->counts, sizeof(host), hipMemcpyDeviceToHost, s));
  SGPO_CHECK_HIP(hipStreamSynchronize(s));
  const bool fits = (size_t)host[0] <= room && !host[2];
  // not fitting: the same kernel with n_needed = 0 appends nothing and releases the slots this batch claimed
  hipLaunchKernelGGL(acq_commit_kernel, dim3((N + 3) / 4), dim3(256), 0, s, a->d, tokens, N, a->n_store,
                     fits ? n_needed : 0, round, a->slot_of, a->flag, a->rank, a->counts + 3);
  SGPO_CHECK_HIP(hipGetLastError());
  int n_tomb = 0;
  SGPO_CHECK_HIP(hipMemcpyAsync(&n_tomb, a->counts + 3, sizeof(int), hipMemcpyDeviceToHost, s));
  SGPO_CHECK_HIP(hipStreamSynchronize(s));
  if (host[2]) {
    SGPO_CHECK_HIP(hipMemsetAsync(a->counts + 2, 0, 4, s));
    sgpo_set_error("acq_update: hash table full (table %zu slots)", a->table);
    return SGPO_ERR_WORKSPACE;
  }
  if (!fits) {
    sgpo_set_error("acq_update: %d rows to append, store holds %zu of %zu", host[0], (size_t)a->n_store, a->capacity);
    return SGPO_ERR_WORKSPACE;
  }
  a->n_store += (uint32_t)host[0];
  *n_kept = host[0];
  if (n_admissible) *n_admissible = host[1];
  // live slots never exceed half the table (sgpo_acq_create); tombstones beyond an eighth of it are swept out
  if ((size_t)n_tomb > a->table / 8) return acq_rebuild(a, s);
  return SGPO_OK;
}

extern "C" int sgpo_acq_read(const sgpo_acq* a, long long first, long long count, int64_t* out_tokens, int* out_rounds,
                             void* stream) {
  SGPO_REQUIRE(a && first >= 0 && count >= 0 && (size_t)(first + count) <= a->n_store, "acq_read: range outside the store");
  if (count == 0) return SGPO_OK;
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)count * a->d.L;
  if (out_tokens)
    hipLaunchKernelGGL(acq_read_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       a->d.store + (size_t)first * a->d.L, n, out_tokens);
  if (out_rounds)
    SGPO_CHECK_HIP(hipMemcpyAsync(out_rounds, a->d.store_round + first, (size_t)count * 4, hipMemcpyDeviceToDevice, s));
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}

extern "C" int sgpo_landscape_lookup(const float* table, size_t table_len, const int64_t* tokens, int N, int L,
                                     const int* residues, int n_res, int n_std, float* out, int* bad_flag, void* stream) {
  SGPO_REQUIRE(table && residues && n_res > 0 && n_std > 0, "landscape_lookup: bad arguments");
  size_t need = 1;
  for (int k = 0; k < n_res; ++k) need *= (size_t)n_std;
  SGPO_REQUIRE(need == table_len, "landscape_lookup: table holds %zu entries, %d^%d = %zu expected", table_len, n_std, n_res, need);
  if (N <= 0) return SGPO_OK;
  hipLaunchKernelGGL(landscape_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, tokens, N, L,
                     residues, n_res, n_std, out, bad_flag);
  SGPO_CHECK_HIP(hipGetLastError());
  return SGPO_OK;
}
