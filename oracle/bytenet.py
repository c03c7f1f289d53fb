"""ORACLE (test infrastructure, not product): CPU fp32 restatement of
`evodiff.model.ByteNetLMTime` -- PARITY UNPINNED.

The module is third party (packages `evodiff` / `sequence_models`, unpinned in the
reference's environment.yaml:20-22) and its source is not under /root/reference, so this
restates the published architecture with the hyper-parameters of
configs/pretrain_model/d3pm.yaml:12-27 and the call site models/d3pm.py:34-37.  No reference
test or fixture pins it; device results are compared against THIS restatement only.
"""
import math

import torch
import torch.nn.functional as F


def positional_encoding(t, d_embedding):
    """PositionalEncoding1D: pe[pos, 0::2] = sin(pos * div), pe[pos, 1::2] = cos(pos * div)."""
    div = torch.exp(torch.arange(0, d_embedding, 2).float() * (-math.log(10000.0) / d_embedding))
    pe = torch.zeros(t.shape[0], d_embedding)
    pe[:, 0::2] = torch.sin(t.float()[:, None] * div)
    pe[:, 1::2] = torch.cos(t.float()[:, None] * div)
    return pe


def n_layers_of(sd):
    return 1 + max(int(k.split('.')[2]) for k in sd if k.startswith('embedder.layers.'))


def dilations(n_layers, r=128):
    log2 = int(math.log2(r)) + 1
    return [2 ** (n % log2) for n in range(n_layers)]


def pff(x, w, b):
    return F.linear(x, w.reshape(w.shape[0], -1), b)


def bytenet_forward(sd, x, t, r=128, return_hidden=False):
    """x (B,L) int64, t (B,) int64 -> logits (B,L,n_tokens)."""
    d_emb = sd['embedder.embedder.weight'].shape[1]
    e = sd['embedder.embedder.weight'][x] + positional_encoding(t, d_emb)[:, None, :]
    h = pff(e, sd['embedder.up_embedder.conv.weight'], sd['embedder.up_embedder.conv.bias'])
    nl = n_layers_of(sd)
    for n, dil in enumerate(dilations(nl, r)):
        p = f'embedder.layers.{n}.'
        a = F.gelu(F.layer_norm(h, [h.shape[-1]], sd[p + 'sequence1.0.weight'], sd[p + 'sequence1.0.bias']))
        a = pff(a, sd[p + 'sequence1.2.conv.weight'], sd[p + 'sequence1.2.conv.bias'])
        a = F.gelu(F.layer_norm(a, [a.shape[-1]], sd[p + 'sequence1.3.weight'], sd[p + 'sequence1.3.bias']))
        k = sd[p + 'conv.weight'].shape[-1]
        a = F.conv1d(a.transpose(1, 2), sd[p + 'conv.weight'], sd[p + 'conv.bias'], dilation=dil,
                     padding=dil * (k - 1) // 2).transpose(1, 2)
        a = F.gelu(F.layer_norm(a, [a.shape[-1]], sd[p + 'sequence2.0.weight'], sd[p + 'sequence2.0.bias']))
        a = pff(a, sd[p + 'sequence2.2.conv.weight'], sd[p + 'sequence2.2.conv.bias'])
        h = h + a
    logits = pff(h, sd['decoder.conv.weight'], sd['decoder.conv.bias'])
    return (logits, h) if return_hidden else logits
