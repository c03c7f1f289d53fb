// Internal launch prototypes for the DiT denoiser kernels.
#pragma once
#include "common.h"

int launch_ln_mod(const float* x, const float* w, const float* shift, const float* scale, int mod_stride,
                  bf16_t* out, int M, int L, int B, int H, hipStream_t stream);
int launch_attention(const bf16_t* qkv, bf16_t* out, int B, int L, int n_heads, int head_dim, hipStream_t stream);
int launch_embed(const int64_t* tok, const float* emb, float* x, int M, int H, int vocab, hipStream_t stream);
int launch_embed_ln_mod(const int64_t* tok, const float* emb, int vocab, float* x, const float* w, const float* shift,
                        const float* scale, int mod_stride, bf16_t* out, int M, int L, int B, int H, hipStream_t stream);
int launch_rope_table(float* cs, float* sn, int max_len, int half, hipStream_t stream);
int launch_conditioning(const float* sigma, int n_sigma, const float* w1, const float* b1, const float* w2,
                        const float* b2, int freq_dim, int cond_dim, float* c_out, const float* wm,
                        const float* bm, int n_out, float* mod, hipStream_t stream);
int launch_f32_to_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t stream);
