/* libsgpo_hip.so -- C ABI of the MI355X-native SGPO guided-sampling hot path.
 *
 * Every entry point replaces a piece of the reference's PyTorch sampling path
 * (citations are file:line into the reference repo, ezirayw/SGPO).  The
 * reference has no FFI of its own: its "plugin API" is the Python duck-type
 * `GenerativeModel` / `Algo` (models/base.py:4-18, sampling/base.py:5-72).  The
 * host mirror in sgpo_amd/dropin/ keeps those Python signatures and calls the
 * functions below through ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; all tensor pointers are DEVICE pointers
 *     unless a parameter says "host".
 *   - token tensors are int64 [B][L] row-major (torch.long), logits/probabilities
 *     fp32 [B][L][K] row-major.
 *   - the caller owns every buffer (including workspaces); the library owns only
 *     the opaque model handles created by *_create.
 *   - calls are asynchronous on `stream` (a hipStream_t passed as void*); no
 *     internal synchronisation, no allocation outside *_create / sgpo_init.
 *   - return 0 on success, negative on error; sgpo_last_error() gives the text.
 *     Nothing throws across the ABI.
 *   - noise: noise_mode 0 = in-kernel Philox4x32-10 keyed by (seed, call offset,
 *     GLOBAL element index; in sgpo_reverse_step_mdlm: GLOBAL row / 4 and class,
 *     word = row % 4) so results are invariant to how a batch is sharded over
 *     ranks (row_offset = first global row of this shard); noise_mode 1/2 = the
 *     caller passes the noise tensor (parity runs).
 */
#ifndef SGPO_HIP_H
#define SGPO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGPO_NOISE_PHILOX 0
#define SGPO_NOISE_EXPONENTIAL 1 /* noise = Exp(1) draws e; sample = argmax(p / e)  (torch.multinomial) */
#define SGPO_NOISE_UNIFORM 2     /* noise = U[0,1) draws u; sample = argmax(p / (1e-10 - log(u + 1e-10))) */
#define SGPO_NOISE_NONE 3        /* no noise: plain argmax (the "noise removal" step, mdlm_diffusion.py:692-699) */

#define SGPO_STEP_SOFT 0    /* write q (unnormalised soft reverse probabilities), no sampling */
#define SGPO_STEP_SOFT_CG 1 /* classifier-guidance step: reweight, normalise, multinomial */
#define SGPO_STEP_HARD 2    /* p_sample(hard=True): _sample_categorical + carry-over unmasking */

typedef struct sgpo_dit sgpo_dit;
typedef struct sgpo_mlp sgpo_mlp;
typedef struct sgpo_bytenet sgpo_bytenet;

typedef struct {
  int vocab;      /* 31 */
  int hidden;     /* 512 */
  int cond_dim;   /* 128 */
  int freq_dim;   /* 256 */
  int n_blocks;   /* 8 */
  int n_heads;    /* 8 (head_dim must be 64) */
  int mlp_hidden; /* 2048 */
  int max_len;    /* RoPE table length (config model.length, 1024) */
} sgpo_dit_config;

/* ---- library ------------------------------------------------------------ */
/* Binds the calling process to `device`.  ONE DEVICE PER PROCESS (one rank per GPU, as torch.distributed.run starts
 * them): launch settings that depend on the device (CU count, tile rule, dynamic-LDS attributes) are cached per
 * process, so a later sgpo_init with another device index fails with SGPO_ERR_ARG instead of running with the first
 * device's settings.  Calling it again with the same index is harmless. */
int sgpo_init(int device);
const char* sgpo_last_error(void);    /* thread-local message of the last failing call */
int sgpo_version(void);

/* ---- MDLM denoiser: DIT.forward (dit.py:375-418), blocks dit.py:244-288 ------ */
/* blob: fp32 parameters (device pointer) concatenated in state_dict order:
 *   vocab_embed.embedding, sigma_map.mlp.0.{weight,bias}, sigma_map.mlp.2.{weight,bias},
 *   per block: norm1.weight, attn_qkv.weight, attn_out.weight, norm2.weight,
 *              mlp.0.{weight,bias}, mlp.2.{weight,bias}, adaLN_modulation.{weight,bias},
 *   output_layer: norm_final.weight, linear.{weight,bias}, adaLN_modulation.{weight,bias}.
 * The library keeps its own bf16/fp32 copies; the blob may be freed afterwards. */
size_t sgpo_dit_param_count(const sgpo_dit_config* cfg);
int sgpo_dit_create(const sgpo_dit_config* cfg, const float* blob, size_t n_floats, sgpo_dit** out);
void sgpo_dit_destroy(sgpo_dit* m);
size_t sgpo_dit_workspace_bytes(const sgpo_dit* m, int B, int L);
/* tokens [B][L]; sigma [n_sigma] (n_sigma = 1 broadcasts, NULL = sigma 0, i.e.
 * time_conditioning False, mdlm_diffusion.py:371-372); x_emb (optional, [B][L][hidden] fp32)
 * bypasses the blocks (dit.py:397-410); hidden_out (optional) receives the last hidden state
 * (return_hidden_states, dit.py:407-408); logits [B][L][vocab] fp32. */
int sgpo_dit_forward(sgpo_dit* m, const int64_t* tokens, const float* sigma, int n_sigma, int B, int L,
                     const float* x_emb, float* logits, float* hidden_out, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ---- fused reverse step, MDLM absorbing state ----------------------------------
 * Covers MDLMDiffusion._subs_parameterization (mdlm_diffusion.py:325-341),
 * MDLM.p_sample (models/mdlm.py:55-98), the classifier-guidance reweighting and
 * torch.multinomial of Classifier_Guidance_Inpaint.inference (sampling/cls_guidance.py:38-52),
 * the ratio clamp/exp of get_guided_rates (models/mdlm.py:399-409) and
 * _sample_categorical (mdlm_diffusion.py:69-73).
 *   log_ratio: [B][n_slots][K] raw log-ratios f(x with x_d<-s) - f(x) (NULL or guide_temp 0 = unguided);
 *   slot_of_pos: [L] slot index of each position or -1 (NULL = identity, n_slots = L);
 *   move_chance_t/s: scalars of models/mdlm.py:67-68 (host computes them, incl. defect D2);
 *   noise: [B][L][K] for noise_mode 1/2;  probs_out (optional) [B][L][K];
 *   nan_flag (optional, device int): bit 0 is set if a guided rate is NaN (models/mdlm.py:410-411), bit 1 if a
 *   row of guided rates sums to 0, inf or NaN (torch.multinomial raises there, cls_guidance.py:52). */
int sgpo_reverse_step_mdlm(const float* logits, const int64_t* x, const float* log_ratio,
                           const int* slot_of_pos, int n_slots, int B, int L, int K, int mask_index,
                           float move_chance_t, float move_chance_s, float guide_temp, int step_mode,
                           int noise_mode, const float* noise, uint64_t seed, uint64_t offset,
                           uint64_t row_offset, int64_t* x_out, float* probs_out, int* nan_flag,
                           void* stream);

/* The same step with the Philox key in DEVICE memory, for callers that record the sampling loop into a HIP graph and
 * replay it with a new seed (no counterpart in the reference, whose noise comes from torch's global generator):
 * key_dev[0] = seed, key_dev[1] = first global row of this shard (rows, i.e. sequences x L), key_dev[2] = base added to
 * `offset` (the call index inside the loop), key_dev[3] unused.  In-kernel Philox only; other arguments as above. */
int sgpo_reverse_step_mdlm_devkey(const float* logits, const int64_t* x, const float* log_ratio,
                                  const int* slot_of_pos, int n_slots, int B, int L, int K, int mask_index,
                                  float move_chance_t, float move_chance_s, float guide_temp, int step_mode,
                                  uint64_t offset, const uint64_t* key_dev, int64_t* x_out, float* probs_out,
                                  int* nan_flag, void* stream);

/* UDLM variant: _d3pm_parameterization + _compute_posterior (mdlm_diffusion.py:343-348, 392-419)
 * + models/mdlm.py:87-95.  step_mode SOFT or HARD. */
int sgpo_reverse_step_udlm(const float* logits, const int64_t* x, int B, int L, int K,
                           float move_chance_t, float move_chance_s, int step_mode, int noise_mode,
                           const float* noise, uint64_t seed, uint64_t offset, uint64_t row_offset,
                           int64_t* x_out, float* probs_out, void* stream);

/* UDLM under Classifier_Guidance_Inpaint.inference (sampling/cls_guidance.py:34-52 on a `diffusion: uniform` net,
 * configs/model/udlm.yaml, scripts/run_prior_sample.sh:12): the posterior above, then -1 at the current token,
 * the clamp/exp reweighting of get_guided_rates (models/mdlm.py:399-411), +1, clamp(min=0), normalise and
 * torch.multinomial.  log_ratio [B][n_slots][K] holds every jump ratio (the uniform prior has no zero-rate
 * columns); NULL or guide_temp 0 = unguided (prior_sample.py:69-76).  Arguments as sgpo_reverse_step_mdlm. */
int sgpo_guided_step_udlm(const float* logits, const int64_t* x, const float* log_ratio, const int* slot_of_pos,
                          int n_slots, int B, int L, int K, float move_chance_t, float move_chance_s,
                          float guide_temp, int noise_mode, const float* noise, uint64_t seed, uint64_t offset,
                          uint64_t row_offset, int64_t* x_out, float* probs_out, int* nan_flag, void* stream);

/* q_xt, uniform branch (mdlm_diffusion.py:593-601) under MDLM.q_sample (models/mdlm.py:104-109): each position
 * is replaced w.p. move_chance[b] by a uniform token of [0, vocab).  noise_mode 2: noise [B][L] uniforms and
 * noise_tokens [B][L] int64 (the reference's torch.rand and torch.randint draws). */
int sgpo_q_sample_udlm(const int64_t* x, const float* move_chance, int B, int L, int vocab, int noise_mode,
                       const float* noise, const int64_t* noise_tokens, uint64_t seed, uint64_t offset,
                       uint64_t row_offset, int64_t* x_out, void* stream);

/* MDLM.q_sample -> q_xt (models/mdlm.py:104-109, mdlm_diffusion.py:585-597): re-mask each
 * position w.p. move_chance[b].  noise: [B][L] uniforms for noise_mode 2. */
int sgpo_q_sample_mdlm(const int64_t* x, const float* move_chance, int B, int L, int mask_index,
                       int noise_mode, const float* noise, uint64_t seed, uint64_t offset,
                       uint64_t row_offset, int64_t* x_out, void* stream);

/* Categorical sampler on its own: rows x K probabilities (need not be normalised). */
int sgpo_categorical_sample(const float* probs, int64_t rows, int K, int noise_mode, const float* noise,
                            uint64_t seed, uint64_t offset, uint64_t row_offset, int64_t* out, void* stream);

/* ---- value function: MLPModel (problem/protein.py:58-121) --------------------------
 * blob (fp32, device): time_embedder.0.W, time_embedder.1.{weight,bias}, embedder.{weight,bias},
 * mlp.0.{weight,bias}, mlp.2.{weight,bias}, mlp.4.{weight,bias}, cls_head.0.{weight,bias},
 * cls_head.2.{weight,bias}.  read_pos (host, [n_read]): the sequence positions the model reads
 * (`seq[:, residues]`, problem/protein.py:108-109 -- the raw config list, defect D3; all
 * positions when residues is None). */
int sgpo_mlp_create(const float* blob, size_t n_floats, int n_read, const int* read_pos, int alphabet,
                    int hidden, int time_conditioned, sgpo_mlp** out);
void sgpo_mlp_destroy(sgpo_mlp* m);
/* dst <- the weights of src (same architecture, same read positions), in place and asynchronous on `stream`: launches
 * recorded with dst (a HIP graph of stacked Thompson draws, iterativeBO.py:204-237) see the new ensemble member on their
 * next replay.  No counterpart in the reference (`update_model(member)` there swaps Python objects). */
int sgpo_mlp_copy(sgpo_mlp* dst, const sgpo_mlp* src, void* stream);
/* forward(seq, t): x [B][L], t [B] (NULL = zeros, log_likelihood of problem/protein.py:37-38) -> out [B] */
int sgpo_mlp_forward(sgpo_mlp* m, const int64_t* x, const float* t, int B, int L, float* out, void* stream);
/* Exact guidance log-ratios of get_guided_rates (models/mdlm.py:360-375, models/d3pm.py:146-161)
 * in factored form: log_ratio [B][n_read][S] = f(x with x[read_pos[i]] <- s, t) - f(x, t).
 * workspace: sgpo_mlp_workspace_bytes(B). */
size_t sgpo_mlp_workspace_bytes(const sgpo_mlp* m, int B);
int sgpo_guidance_ratio_mlp(sgpo_mlp* m, const int64_t* x, float t, int B, int L, int S, float* log_ratio,
                            void* workspace, size_t workspace_bytes, void* stream);
/* The same for absorbing-state diffusion (MDLM), computing only the ratios the guided rates can depend on.
 * At a position whose token is not `mask_index` the subs-parameterised prior of MDLM.p_sample
 * (models/mdlm.py:164-182: carry-over one-hot plus move_chance_s in the mask column) is exactly 0 for
 * every target except the token itself and the mask, so cls_guidance.py:38-52 multiplies every other
 * ratio by 0: those entries are written as 0 here instead of being evaluated (entries at masked positions,
 * and the mask column everywhere, are bit-identical to sgpo_guidance_ratio_mlp).  Same workspace. */
int sgpo_guidance_ratio_mlp_absorbing(sgpo_mlp* m, const int64_t* x, float t, int B, int L, int S, int mask_index,
                                      float* log_ratio, void* workspace, size_t workspace_bytes, void* stream);
/* One guided step (cls_guidance.py:34-52) for SEVERAL sampler calls stacked in one batch, e.g. the Thompson draws of
 * iterativeBO.py:204-237 (`update_model(member); inference(10)` per draw): rows [row_lo[g], row_lo[g+1]) belong to
 * draw g with value function members[g] (NULL: unguided) and Philox key (seeds[g], offsets[g], row_offsets[g] -- the
 * arguments sgpo_reverse_step_mdlm would get for that draw alone).  The kernels of sgpo_guidance_ratio_mlp_absorbing
 * and of sgpo_reverse_step_mdlm (SGPO_STEP_SOFT_CG, in-kernel Philox) run once for up to 16 draws at a time; inside a
 * draw every index is that of its own call, so every draw gets the tokens of its own call, bit for bit.  The value
 * functions must read the same positions.  members, row_lo (n_groups + 1, row_lo[0] = 0), seeds, offsets, row_offsets
 * are HOST arrays.  log_ratio / workspace: device scratch for all B = row_lo[n_groups] rows, B * n_read * S floats and
 * sgpo_mlp_workspace_bytes(m, B) bytes.  logits [B][L][K], x and x_out [B][L]. */
int sgpo_guided_step_groups_mdlm(int n_groups, sgpo_mlp* const* members, const int* row_lo, const float* logits,
                                 const int64_t* x, int L, int K, int S, float t, int mask_index, float move_chance_t,
                                 float move_chance_s, float guide_temp, const uint64_t* seeds, const uint64_t* offsets,
                                 const uint64_t* row_offsets, const int* slot_of_pos, float* log_ratio,
                                 size_t log_ratio_bytes, void* workspace, size_t workspace_bytes, int64_t* x_out,
                                 int* nan_flag, void* stream);
/* The same with the draws' Philox keys in DEVICE memory (HIP-graph replay with new seeds; see
 * sgpo_reverse_step_mdlm_devkey): keys_dev [n_groups][4] = seed, first global row, offset base, unused; `offset` is
 * the call index inside the loop, the same for every draw.  At most 1024 draws. */
int sgpo_guided_step_groups_mdlm_devkey(int n_groups, sgpo_mlp* const* members, const int* row_lo, const float* logits,
                                        const int64_t* x, int L, int K, int S, float t, int mask_index,
                                        float move_chance_t, float move_chance_s, float guide_temp, uint64_t offset,
                                        const uint64_t* keys_dev, const int* slot_of_pos, float* log_ratio,
                                        size_t log_ratio_bytes, void* workspace, size_t workspace_bytes, int64_t* x_out,
                                        int* nan_flag, void* stream);
/* sgpo_guidance_ratio_mlp_absorbing for callers that cannot allocate.  No counterpart in the reference: it is for
 * code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc, which have no way to
 * provide the workspace.  Same arguments minus the workspace, same kernels, same bits in log_ratio.  The
 * workspace (sgpo_mlp_workspace_bytes(m, B) bytes) is a device array inside the library, at most
 * SGPO_GUIDANCE_NOALLOC_MAX_BYTES per GPU: a call that needs more (e.g. more than about 5 600 sequences with 119 read positions)
 * returns an error naming the sizes.  The call allocates nothing (no hipMalloc/hipFree, no host allocation with
 * per-kernel timing off).  Every call on a GPU shares that GPU's array: calls that overlap on one GPU
 * (different streams or threads) race; calls on one stream are ordered.  Nothing in the array has to start at
 * zero -- each call writes every word it later reads. */
#define SGPO_GUIDANCE_NOALLOC_MAX_BYTES (4u << 20)
size_t sgpo_guidance_noalloc_max_bytes(void);
int sgpo_guidance_ratio_mlp_absorbing_noalloc(sgpo_mlp* m, const int64_t* x, float t, int B, int L, int S,
                                              int mask_index, float* log_ratio, void* stream);

/* ---- fitness-oracle ensemble: oracle/inference_oracle.py:19-72 ---------
 * OracleModel (oracle/train_oracle.py:61-74) = Linear(20*n_res, hidden) -> ReLU -> Linear(hidden, 1) on the
 * one-hot of the 20-letter alphabet 'ACDEFGHIKLMNPQRSTVWY' (train_oracle.py:39-48), n_members of them.
 * blob (fp32, device), per member in order: fc1.weight [hidden][20*n_res], fc1.bias [hidden],
 * fc2.weight [hidden], fc2.bias [1].
 * forward: tokens [N][n_res] int64 (index into the 20-letter alphabet = the tokenizer id for these letters);
 * parent_tokens [n_res] (NULL: impose_penalty=False); out[n] = mean over members of
 * max(pred * (hd <= cutoff ? 1 : rate^(hd - cutoff)), 0) in fp64, hd = Hamming distance to the parent
 * (inference_oracle.py:10-17,47-68).  *bad_token_flag (device int, may be NULL) is set when a token lies
 * outside [0, 20) -- the reference raises ValueError there.  workspace: sgpo_fitness_workspace_bytes(N). */
typedef struct sgpo_fitness sgpo_fitness;
int sgpo_fitness_create(const float* blob, size_t n_floats, int n_members, int n_res, int hidden, sgpo_fitness** out);
void sgpo_fitness_destroy(sgpo_fitness* m);
size_t sgpo_fitness_workspace_bytes(const sgpo_fitness* m, int N);
int sgpo_fitness_forward(sgpo_fitness* m, const int64_t* tokens, int N, const int64_t* parent_tokens, int cutoff,
                         double rate, double* out, int* bad_token_flag, void* workspace, size_t workspace_bytes,
                         void* stream);

/* ---- batch hand-over between sampler and fitness oracle on token tensors ----
 * Token-level Algo.project_sequences (sampling/base.py:40-63).  tokens [N][L] int64, repaired in place.
 *  1. a token >= n_std (a special character, base.py:45-47) becomes floor(u * n_std) with
 *     u = u_special[n][token - n_std]: one draw per distinct special token of a sequence, since
 *     str.replace there replaces every occurrence at once.  u_special [N][n_special] fp32 in [0,1), or NULL
 *     (then a special token sets bit 1 of *bad_flag and is left alone).
 *  2. n_max_mutations >= 0 (the caller applies base.py:50's `n_residues > 4 and n_max_mutations is not None`):
 *     of the positions that differ from full_seq [L], the n_max_mutations with the smallest keys[n][j]
 *     (ties: smaller j) stay, the others are reverted to full_seq -- np.random.choice(mutated, k,
 *     replace=False) of base.py:53-61 with the randomness handed in as iid keys [N][L] fp32.
 * A token outside [0, n_std + n_special) sets bit 0 of *bad_flag (device int, may be NULL).  L <= 1024. */
int sgpo_project_tokens(int64_t* tokens, const int64_t* full_seq, int N, int L, int n_std, int n_special,
                        const float* u_special, int n_max_mutations, const float* keys, int* bad_flag, void* stream);
/* The sequence store of ProteinPredictorDataset (dataset/protein.py:41-44,153-183) as a hash set in HBM.
 * capacity: rows the store can hold; max_batch: largest N of one update; capacity + max_batch < 2^24. */
typedef struct sgpo_acq sgpo_acq;
int sgpo_acq_create(int L, size_t capacity, int max_batch, sgpo_acq** out);
void sgpo_acq_destroy(sgpo_acq* a);
long long sgpo_acq_count(const sgpo_acq* a);
/* update_data's filter (dataset/protein.py:164-183) on tokens [N][L] int64 (values < 256).
 * scope 0: keep every row (the reference's path when the store is empty, at the first call of a round without
 *          BO, or with unique_only=False); 1: unique rows not yet in the store (BO=True, protein.py:168-170);
 *          2: unique rows not appended during `round` (protein.py:171-173).
 * Rows are kept in batch order (the reference's list(set(...)) order depends on the interpreter's string hash
 * seed), the first n_needed of them (protein.py:178-179) are appended to the store with their round.
 * kept_idx [N] int64 (device): batch rows appended, in order; *n_kept (host): how many; *n_admissible (host,
 * may be NULL): how many passed the filter before truncation.  Synchronises the stream. */
int sgpo_acq_update(sgpo_acq* a, const int64_t* tokens, int N, int n_needed, int scope, int round,
                    int64_t* kept_idx, int* n_kept, int* n_admissible, void* stream);
/* rows [first, first+count) of the store: out_tokens [count][L] int64 and/or out_rounds [count] int32 (device). */
int sgpo_acq_read(const sgpo_acq* a, long long first, long long count, int64_t* out_tokens, int* out_rounds,
                  void* stream);
/* update_data's landscape branch (dataset/protein.py:188-197): table [n_std^n_res] fp32 indexed by the tokens at
 * residues [n_res] (device int32), most significant first; combos the data frame lacks hold 0 (protein.py:193). */
int sgpo_landscape_lookup(const float* table, size_t table_len, const int64_t* tokens, int N, int L,
                          const int* residues, int n_res, int n_std, float* out, int* bad_flag, void* stream);

/* ---- value-function ensemble training: training/train_classifier.py:7-64 ----
 * Adam (torch.optim.Adam defaults, train_classifier.py:23) on MSELoss (train_classifier.py:24) of MLPModel
 * (problem/protein.py:58-121), n_members models advanced in lockstep: one call = one optimiser step of
 * train_step (train_classifier.py:7-19) for every member on that member's own batch.
 * blobs (fp32, device) [n_members][P], P = sgpo_mlp_trainer_param_count(...), each in the order of sgpo_mlp_create;
 * read_pos as there (HOST int32).  The caller draws t, applies q_sample and project_fn (train_classifier.py:13-14)
 * and passes the result: x [n_members][B][L] int64, t [n_members][B] fp32 (NULL when not time-conditioned),
 * y [n_members][B] fp32, 1 <= B <= 64.  loss_out [n_members] (device) = the step's MSE before the update.
 * GaussianFourierProjection.W is not trained (requires_grad=False, protein.py:52).  Every reduction runs in a
 * fixed order: the same inputs give the same bits. */
typedef struct sgpo_mlp_trainer sgpo_mlp_trainer;
size_t sgpo_mlp_trainer_param_count(int n_read, int alphabet, int time_conditioned);
int sgpo_mlp_trainer_create(const float* blobs, int n_members, int n_read, const int* read_pos, int alphabet,
                            int hidden, int time_conditioned, sgpo_mlp_trainer** out);
void sgpo_mlp_trainer_destroy(sgpo_mlp_trainer* t);
int sgpo_mlp_trainer_step(sgpo_mlp_trainer* t, const int64_t* x, const float* tt, const float* y, int B, int L,
                          double lr, double beta1, double beta2, double eps, float* loss_out, void* stream);
/* n_steps consecutive optimiser steps in one call: batch j of every member at x + j*[n_members][B][L] (likewise t, y),
 * its losses at loss_out + j*[n_members].  Same kernels and bits as n_steps calls of sgpo_mlp_trainer_step (the batches of
 * an epoch of train_classifier.py:26-41 are known up front); saves the caller's per-step overhead (0.32 -> 0.25 ms). */
int sgpo_mlp_trainer_steps(sgpo_mlp_trainer* t, const int64_t* x, const float* tt, const float* y, int n_steps, int B, int L,
                           double lr, double beta1, double beta2, double eps, float* loss_out, void* stream);
int sgpo_mlp_trainer_read(const sgpo_mlp_trainer* t, float* blobs_out, void* stream);

/* ---- NOS step: MDLM._nos_denoise (models/mdlm.py:184-338) with the DITClassifier head on x_emb
 * (dit.py:494-515: mean-pool over L + Linear(hidden, 1); w_cls = its weight, [hidden] fp32).
 * One backbone pass, then num_nos_steps closed-form gradient + Adagrad updates of delta
 * (B x L x hidden; held in registers over all inner steps by ONE kernel launch, written out only
 * through delta_out) on the final layer only, then the guided reverse step
 * (_sample_categorical + carry-over).  kl_batch: the B of KLDivLoss(reduction='batchmean')
 * (models/mdlm.py:254,297) -- pass the GLOBAL per-call batch when a batch is sharded over ranks.
 * noise: [B][L][vocab] uniforms for noise_mode 2.  probs_out (optional) [B][L][vocab] = guided_probs;
 * delta_out (optional) [B][L][hidden]; hidden_in (optional) [B][L][hidden]: use this last hidden
 * state instead of running the blocks.  Absorbing-state + subs parameterisation only. */
size_t sgpo_nos_workspace_bytes(const sgpo_dit* m, int B, int L);
int sgpo_nos_denoise(sgpo_dit* m, const int64_t* xt, const float* sigma, int n_sigma, int B, int L,
                     const float* w_cls, int num_nos_steps, float nos_step_size, float nos_stability_coef,
                     float move_chance_t, float move_chance_s, int kl_batch, int mask_index, int noise_mode,
                     const float* noise, uint64_t seed, uint64_t offset, uint64_t row_offset,
                     const float* hidden_in, int64_t* xs_out, float* probs_out, float* delta_out,
                     void* nos_workspace, size_t nos_workspace_bytes, void* dit_workspace,
                     size_t dit_workspace_bytes, void* stream);

/* ---- D3PM denoiser: evodiff.model.ByteNetLMTime (third party; call site models/d3pm.py:34-37,
 * hyper-parameters configs/pretrain_model/d3pm.yaml:12-27) -- restated, parity unpinned ----------
 * blob (fp32, device), state_dict order: embedder.embedder.weight, embedder.up_embedder.conv.{weight,bias},
 * per layer n: sequence1.0.{weight,bias}, sequence1.2.conv.{weight,bias}, sequence1.3.{weight,bias},
 * conv.{weight,bias} ([out][in][tap]), sequence2.0.{weight,bias}, sequence2.2.conv.{weight,bias};
 * decoder.conv.{weight,bias}. */
typedef struct {
  int n_tokens;    /* 31 */
  int d_embedding; /* 8 */
  int d_model;     /* 1024 */
  int d_h;         /* 512 (slim) */
  int n_layers;    /* 16 */
  int kernel_size; /* 5 */
  int r;           /* 128: dilation of layer n = 2^(n mod (log2(r)+1)) */
} sgpo_bytenet_config;
size_t sgpo_bytenet_param_count(const sgpo_bytenet_config* cfg);
int sgpo_bytenet_create(const sgpo_bytenet_config* cfg, const float* blob, size_t n_floats, sgpo_bytenet** out);
void sgpo_bytenet_destroy(sgpo_bytenet* m);
size_t sgpo_bytenet_workspace_bytes(const sgpo_bytenet* m, int B, int L);
/* tokens [B][L]; t [n_t] int64 timesteps (n_t = 1 broadcasts); logits [B][L][n_tokens] fp32;
 * hidden_out optional [B][L][d_model]. */
int sgpo_bytenet_forward(sgpo_bytenet* m, const int64_t* tokens, const int64_t* t, int n_t, int B, int L,
                         float* logits, float* hidden_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- D3PM reverse step: D3PM.p_sample (models/d3pm.py:230-268) -------------------------------
 * logits [N][L][ld_logits] fp32 (first S columns used, pred_mean models/d3pm.py:34-37);
 * delta_q [S][S] fp32 = Q[t] or the analytic e^{dsigma} I + (1-e^{dsigma})/S (models/d3pm.py:239-243);
 * q_bar [S][S] fp32 = Q_bar[t-1] or Q_bar[t_next] (models/d3pm.py:250-253); both chosen by the host.
 * SOFT: probs_out [N][L][S] fp64 (normalised posterior).  HARD: x_out; noise (noise_mode 1) is an
 * fp64 [N][L][S] tensor of Exp(1) draws (torch.multinomial on fp64 probabilities). */
int sgpo_reverse_step_d3pm(const float* logits, int ld_logits, const int64_t* x, int N, int L, int S,
                           const float* delta_q, const float* q_bar, int step_mode, int noise_mode,
                           const void* noise, uint64_t seed, uint64_t offset, uint64_t row_offset,
                           int64_t* x_out, double* probs_out, void* stream);
/* D3PM.q_sample (models/d3pm.py:45-52): t [B] int64; q_bar_all [T][S][S] fp32; noise fp32 [B][L][S] Exp(1). */
int sgpo_q_sample_d3pm(const int64_t* x, const int64_t* t, const float* q_bar_all, int T, int B, int L, int S,
                       int noise_mode, const float* noise, uint64_t seed, uint64_t offset, uint64_t row_offset,
                       int64_t* x_out, void* stream);

/* ---- DAPS Metropolis-Hastings: DAPS.metropolis_hasting + log_ratio (sampling/daps.py:68-114) ----
 * `steps` sequential proposals per sequence in one launch.  residues / res_slot: DEVICE int32
 * [n_res] = mutable positions (0-indexed, sampling/base.py:30) and the read slot of each in the value
 * function (or -1: defect D3 makes some proposals invisible to it).  log_alpha / log_1alpha: the
 * two scalars of log_ratio (daps.py:69-71) from sigma/beta, computed by the host in fp32.
 * Injected noise (noise_mode != 0): r_idx/r_val int32 [steps][N], r_u fp32 [steps][N].
 * x_out must not alias x0hat.  workspace: sgpo_daps_mh_workspace_bytes(). */
size_t sgpo_daps_mh_workspace_bytes(void);
int sgpo_daps_mh(sgpo_mlp* m, const int64_t* x0hat, int N, int L, int S, const int* residues,
                 const int* res_slot, int n_res, float log_alpha, float log_1alpha, float alpha, int steps,
                 int noise_mode, const int* r_idx, const int* r_val, const float* r_u, uint64_t seed,
                 uint64_t offset, uint64_t row_offset, int64_t* x_out, void* workspace,
                 size_t workspace_bytes, void* stream);

/* ---- test / bench access to the building blocks ----------------------------------- */
/* C[M][N] (fp32) = A[M][K] (bf16) * W[N][K]^T (bf16) through the MFMA GEMM kernel. */
int sgpo_gemm_bf16(const void* A, const void* W, float* C, int M, int N, int K, void* stream);
/* test / bench access to the two-workgroups-per-CU GEMM (gemm_q.hip): C bf16 [M][N] = epi(A W^T + bias), A [M][K] and
 * W [N][K] bf16, bias [N] fp32 or NULL; epi 0 = bias only, 3 = tanh-GELU (the fc1 epilogue of dit.py:302-321). */
int sgpo_gemm_q_bf16(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int epi, void* stream);
int sgpo_f32_to_bf16(const float* src, void* dst, size_t n, void* stream);

/* ---- per-kernel timing (HIP events on the launch stream; bench.py's roofline leg) -----
 * No counterpart in the reference (it has no profiling).  enable(1, n)
 * preallocates events for n launches; read() synchronises and returns, per kernel kind,
 * total milliseconds, launch count and total algorithmic work (flops for MFMA kernels,
 * bytes for HBM-bound kernels).  SGPO_PROF_KINDS slots are required. */
#define SGPO_PROF_KINDS 18
int sgpo_prof_enable(int on, int max_launches);
int sgpo_prof_reset(void);
int sgpo_prof_read(int n_kinds, double* total_ms, int64_t* launches, double* work);
const char* sgpo_prof_kind_name(int kind);

#ifdef __cplusplus
}
#endif
#endif /* SGPO_HIP_H */
