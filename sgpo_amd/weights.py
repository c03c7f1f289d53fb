"""Weight containers for the denoisers and value functions on the sampling path.

State-dict key names follow the reference modules so that a real checkpoint's
`state_dict` (prefix stripped) can be fed in unchanged:
  * DiT      -> models/pretraining/model/mdlm/models/dit.py:324-354 (`backbone.*`)
  * MLPModel -> problem/protein.py:58-95
  * ByteNet  -> evodiff.model.ByteNetLMTime (third party; SURVEY.md Appendix A, `network.*`)

There are no pretrained checkpoints offline, so benchmarks and
parity tests use the seeded synthetic initialisers below.  Layers the reference
zero-initialises (adaLN, final linear) are re-drawn N(0, 0.02) so the logits are
non-trivial (SURVEY.md §8d "Synthetic inputs").
"""
import math
import os

import torch


def _lin(g, out_f, in_f, bias=True, prefix=''):
    bound = 1.0 / math.sqrt(in_f)
    d = {prefix + 'weight': (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound}
    if bias:
        d[prefix + 'bias'] = (torch.rand(out_f, generator=g) * 2 - 1) * bound
    return d


def dit_synthetic_state(seed=0, vocab=31, hidden=512, cond_dim=128, n_blocks=8,
                        n_heads=8, mlp_ratio=4, freq_dim=256):
    """Seeded fp32 state dict of the DIT architecture (dit.py:324-354)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    bound = 1.0 / math.sqrt(hidden)
    sd['vocab_embed.embedding'] = (torch.rand(vocab, hidden, generator=g) * 2 - 1) * bound * math.sqrt(3.0) * 8
    sd.update(_lin(g, cond_dim, freq_dim, prefix='sigma_map.mlp.0.'))
    sd.update(_lin(g, cond_dim, cond_dim, prefix='sigma_map.mlp.2.'))
    hd = hidden // n_heads
    sd['rotary_emb.inv_freq'] = 1.0 / (10000 ** (torch.arange(0, hd, 2).float() / hd))
    for i in range(n_blocks):
        p = f'blocks.{i}.'
        sd[p + 'norm1.weight'] = 1.0 + 0.1 * torch.randn(hidden, generator=g)
        sd.update(_lin(g, 3 * hidden, hidden, bias=False, prefix=p + 'attn_qkv.'))
        sd.update(_lin(g, hidden, hidden, bias=False, prefix=p + 'attn_out.'))
        sd[p + 'norm2.weight'] = 1.0 + 0.1 * torch.randn(hidden, generator=g)
        sd.update(_lin(g, mlp_ratio * hidden, hidden, prefix=p + 'mlp.0.'))
        sd.update(_lin(g, hidden, mlp_ratio * hidden, prefix=p + 'mlp.2.'))
        sd[p + 'adaLN_modulation.weight'] = 0.02 * torch.randn(6 * hidden, cond_dim, generator=g)
        # gates centred on 1 so the residual branches contribute (the trained model's do)
        b = 0.02 * torch.randn(6 * hidden, generator=g)
        b[2 * hidden:3 * hidden] += 1.0
        b[5 * hidden:6 * hidden] += 1.0
        sd[p + 'adaLN_modulation.bias'] = b
    sd['output_layer.norm_final.weight'] = 1.0 + 0.1 * torch.randn(hidden, generator=g)
    sd['output_layer.linear.weight'] = 0.05 * torch.randn(vocab, hidden, generator=g)
    sd['output_layer.linear.bias'] = 0.02 * torch.randn(vocab, generator=g)
    sd['output_layer.adaLN_modulation.weight'] = 0.02 * torch.randn(2 * hidden, cond_dim, generator=g)
    sd['output_layer.adaLN_modulation.bias'] = 0.02 * torch.randn(2 * hidden, generator=g)
    return sd


def dit_dims(sd):
    vocab, hidden = sd['vocab_embed.embedding'].shape
    cond_dim, freq_dim = sd['sigma_map.mlp.0.weight'].shape
    n_blocks = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('blocks.'))
    n_heads = hidden // (2 * sd['rotary_emb.inv_freq'].numel())
    mlp_hidden = sd['blocks.0.mlp.0.weight'].shape[0]
    return dict(vocab=vocab, hidden=hidden, cond_dim=cond_dim, freq_dim=freq_dim,
                n_blocks=n_blocks, n_heads=n_heads, mlp_hidden=mlp_hidden)


def mlp_synthetic_state(seed=0, n_residues=119, alphabet=31, hidden=64, time_conditioned=True):
    """Seeded fp32 state dict of MLPModel (problem/protein.py:58-95)."""
    g = torch.Generator().manual_seed(seed)
    sd = {'time_embedder.0.W': torch.randn(hidden // 2, generator=g) * 30.0}
    sd.update(_lin(g, hidden, hidden, prefix='time_embedder.1.'))
    sd.update(_lin(g, hidden, alphabet, prefix='embedder.'))
    in_dim = hidden * (n_residues + 1) if time_conditioned else hidden * n_residues
    sd.update(_lin(g, hidden, in_dim, prefix='mlp.0.'))
    sd.update(_lin(g, hidden, hidden, prefix='mlp.2.'))
    sd.update(_lin(g, hidden, hidden, prefix='mlp.4.'))
    sd.update(_lin(g, hidden, hidden, prefix='cls_head.0.'))
    sd.update(_lin(g, 1, hidden, prefix='cls_head.2.'))
    # spread the outputs a little so guidance ratios are O(1) rather than O(1e-2)
    sd['mlp.0.weight'] = sd['mlp.0.weight'] * 4.0
    return sd


def bytenet_synthetic_state(seed=0, n_tokens=31, d_embedding=8, d_model=1024, n_layers=16,
                            kernel_size=5, slim=True):
    """Seeded fp32 state dict of ByteNetLMTime (parity unpinned)."""
    g = torch.Generator().manual_seed(seed)
    d_h = d_model // 2 if slim else d_model
    sd = {'embedder.embedder.weight': torch.randn(n_tokens, d_embedding, generator=g)}
    sd['embedder.embedder.weight'][28] = 0.0  # padding_idx = mask id
    sd.update(_lin(g, d_model, d_embedding, prefix='embedder.up_embedder.conv.'))
    for n in range(n_layers):
        p = f'embedder.layers.{n}.'
        sd[p + 'sequence1.0.weight'] = 1.0 + 0.1 * torch.randn(d_model, generator=g)
        sd[p + 'sequence1.0.bias'] = 0.1 * torch.randn(d_model, generator=g)
        sd.update(_lin(g, d_h, d_model, prefix=p + 'sequence1.2.conv.'))
        sd[p + 'sequence1.3.weight'] = 1.0 + 0.1 * torch.randn(d_h, generator=g)
        sd[p + 'sequence1.3.bias'] = 0.1 * torch.randn(d_h, generator=g)
        bound = 1.0 / math.sqrt(d_h * kernel_size)
        sd[p + 'conv.weight'] = (torch.rand(d_h, d_h, kernel_size, generator=g) * 2 - 1) * bound
        sd[p + 'conv.bias'] = (torch.rand(d_h, generator=g) * 2 - 1) * bound
        sd[p + 'sequence2.0.weight'] = 1.0 + 0.1 * torch.randn(d_h, generator=g)
        sd[p + 'sequence2.0.bias'] = 0.1 * torch.randn(d_h, generator=g)
        sd.update(_lin(g, d_model, d_h, prefix=p + 'sequence2.2.conv.'))
    sd.update(_lin(g, n_tokens, d_model, prefix='decoder.conv.'))
    return sd


def load_lightning_state(path, prefix):
    """Read `state_dict` from a Lightning .ckpt (local file only) and strip `prefix`.

    Checkpoint/resume hook of SURVEY.md §5: `.ckpt -> flat weight dict`.  Only the non-executing loader
    (`weights_only=True`) is ever used: a full unpickle runs code from the file, and there is no switch that
    turns it on.  A Lightning checkpoint that carries pickled hyper-parameter objects is refused; re-save its
    tensors (`torch.save({'state_dict': ckpt['state_dict']}, path)`) on a machine where the file is trusted.
    """
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    try:
        ck = torch.load(path, map_location='cpu', weights_only=True)
    except Exception as e:
        raise RuntimeError(f'{path}: the non-executing loader refused this checkpoint ({type(e).__name__}: {e}). '
                           f"Re-save its tensors with torch.save({{'state_dict': ckpt['state_dict']}}, path); "
                           f'this library never unpickles a checkpoint.') from e
    sd = ck['state_dict'] if 'state_dict' in ck else ck
    out = {}
    for k, v in sd.items():
        if k.startswith(prefix):
            out[k[len(prefix):]] = v.detach().float().contiguous()
    if not out:
        raise KeyError(f'no keys with prefix {prefix!r} in {path}')
    return out


def oracle_ensemble_synthetic_state(seed=0, n_members=10, n_res=119, hidden=400):
    """Seeded fp32 state dicts of an OracleModel ensemble (oracle/train_oracle.py:61-74): fc1
    (hidden x 20*n_res), fc2 (1 x hidden).  The output bias is shifted per member so that some
    predictions are negative (exercises the max(., 0) of oracle/inference_oracle.py:62)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for m in range(n_members):
        sd = {}
        sd.update(_lin(g, hidden, 20 * n_res, prefix='fc1.'))
        sd.update(_lin(g, 1, hidden, prefix='fc2.'))
        sd['fc1.weight'] = sd['fc1.weight'] * 3.0
        sd['fc2.bias'] = sd['fc2.bias'] + 0.05 * (m % 3 - 1)
        out.append(sd)
    return out
