"""ORACLE (test infrastructure, not product): CPU fp32 restatement of the MDLM
denoiser.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this package.

Follows /root/reference/models/pretraining/model/mdlm/models/dit.py:
  Rotary            :81-115  (+ flash_attn rotate-half RoPE)
  LayerNorm         :126-134 (weight only, fp32)
  TimestepEmbedder  :150-189
  DDiTBlock.forward :244-288 (eval mode: dropout off, defect D6)
  DDitFinalLayer    :302-321
  DIT.forward       :375-418
  DITClassifier.forward(x_emb=) :475-515

Pinned against goldens captured from the reference import
(tests/golden/dit_*.npz, made by tools/refgen/make_goldens.py).
`emulate_bf16=True` rounds GEMM/attention operands and outputs to bf16 at the
points torch autocast does on a GPU (defect D7), to tighten device comparisons.
"""
import math

import torch
import torch.nn.functional as F


def _r(x, on):
    return x.to(torch.bfloat16).to(torch.float32) if on else x


def _linear(x, w, b, bf16):
    y = F.linear(_r(x, bf16), _r(w, bf16), None if b is None else _r(b, bf16))
    return _r(y, bf16)


def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def conditioning(sd, sigma, bf16=False):
    """c = silu(sigma_map(sigma)) (dit.py:385).  sigma_map runs outside autocast -> fp32."""
    freq_dim = sd['sigma_map.mlp.0.weight'].shape[1]
    e = timestep_embedding(sigma, freq_dim)
    h = F.linear(e, sd['sigma_map.mlp.0.weight'], sd['sigma_map.mlp.0.bias'])
    h = F.silu(h)
    h = F.linear(h, sd['sigma_map.mlp.2.weight'], sd['sigma_map.mlp.2.bias'])
    return F.silu(h)


def layer_norm(x, w):
    return F.layer_norm(x.float(), [x.shape[-1]]) * w[None, None, :]


def rope_tables(sd, L):
    inv_freq = sd['rotary_emb.inv_freq']
    t = torch.arange(L, dtype=torch.float32)
    freqs = torch.einsum('i,j->ij', t, inv_freq)
    return freqs.cos(), freqs.sin()  # (L, hd/2)


def block_forward(sd, i, x, c, cos, sin, n_heads, bf16=False):
    p = f'blocks.{i}.'
    B, L, H = x.shape
    hd = H // n_heads
    mod = _linear(c, sd[p + 'adaLN_modulation.weight'], sd[p + 'adaLN_modulation.bias'], bf16)
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = mod[:, None].chunk(6, dim=2)
    x_skip = x
    h = layer_norm(x, sd[p + 'norm1.weight']) * (1 + scale_msa) + shift_msa
    qkv = _linear(h, sd[p + 'attn_qkv.weight'], None, bf16).view(B, L, 3, n_heads, hd)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    c_, s_ = cos[None, :, None, :], sin[None, :, None, :]

    def rot(u):
        u1, u2 = u[..., :hd // 2], u[..., hd // 2:]
        return _r(torch.cat([u1 * c_ - u2 * s_, u1 * s_ + u2 * c_], dim=-1), bf16)

    q, k = rot(q), rot(k)
    att = torch.einsum('blhd,bmhd->bhlm', q, k) / math.sqrt(hd)
    att = att.softmax(dim=-1)
    o = _r(torch.einsum('bhlm,bmhd->blhd', _r(att, bf16), v), bf16).reshape(B, L, H)
    x = x_skip + gate_msa * _linear(o, sd[p + 'attn_out.weight'], None, bf16)
    h = layer_norm(x, sd[p + 'norm2.weight']) * (1 + scale_mlp) + shift_mlp
    h = _linear(h, sd[p + 'mlp.0.weight'], sd[p + 'mlp.0.bias'], bf16)
    h = _r(F.gelu(h, approximate='tanh'), bf16)
    h = _linear(h, sd[p + 'mlp.2.weight'], sd[p + 'mlp.2.bias'], bf16)
    return x + gate_mlp * h


def final_layer(sd, x, c, bf16=False):
    mod = _linear(c, sd['output_layer.adaLN_modulation.weight'], sd['output_layer.adaLN_modulation.bias'], bf16)
    shift, scale = mod[:, None].chunk(2, dim=2)
    h = layer_norm(x, sd['output_layer.norm_final.weight']) * (1 + scale) + shift
    return _linear(h, sd['output_layer.linear.weight'], sd['output_layer.linear.bias'], bf16)


def n_heads_of(sd):
    hidden = sd['vocab_embed.embedding'].shape[1]
    return hidden // (2 * sd['rotary_emb.inv_freq'].numel())


def dit_forward(sd, indices, sigma, x_emb=None, return_hidden_states=False, emulate_bf16=False):
    """DIT.forward (dit.py:375-418).  indices (B,L) int64, sigma (B,) f32 -> logits (B,L,V)."""
    bf16 = emulate_bf16
    c = conditioning(sd, sigma)
    hidden_states = []
    if x_emb is None:
        x = sd['vocab_embed.embedding'][indices]
        hidden_states.append(x)
        cos, sin = rope_tables(sd, x.shape[1])
        nb = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('blocks.'))
        nh = n_heads_of(sd)
        for i in range(nb):
            x = block_forward(sd, i, x, c, cos, sin, nh, bf16)
            hidden_states.append(x)
    else:
        x = x_emb
    logits = final_layer(sd, x, c, bf16)
    if return_hidden_states:
        return logits, hidden_states
    return logits


def dit_classifier_head(w, b, x_emb):
    """DITClassifier.forward with x_emb given and pooling='mean' (dit.py:494-515)."""
    return F.linear(x_emb.mean(dim=1), w, b)
