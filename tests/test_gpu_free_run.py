"""north_star's third correctness check at the real dims, joined to row f-1: "the final runs
agree in per-position amino-acid marginals and oracle-fitness mean/diversity within stated bounds".

Reference side (tests/golden/free_run_real.npz, written by tools/refgen/make_goldens.py gen_free_run): 64 sequences of
the reference's own guided free run -- MDLM + Classifier_Guidance_Inpaint at the real DiT dims on CreiLOV, 127 steps,
temperature 0.064, detokenize=True -> project_sequences -- scored by the reference's inference_oracle on its SHIPPED
CreiLOV ensemble.  Device side: 2048 sequences through the drop-in (Philox noise), repaired by the drop-in's
project_sequences, scored by `fitness.inference_oracle` (sgpo_fitness_forward) on the same ensemble weights
(oracle/_ref/CreiLOV_ens, exported by __graft_entry__.build() from the reference checkout; not in git).

Stated bounds: a free run of 64 cannot be compared value by value, so every statistic of the reference's 64 is placed
in the distribution of the same statistic over random 64-subsets of the device's 2048 samples (200 subsets): it must
fall inside the central 99.5 % band widened by 20 % of the band's width.  Statistics: mean per-position total
variation to the device's pooled marginals, mean per-position Shannon entropy (analysis.ipynb cell 0), mean and
standard deviation of the fitness, mean value-function score."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu

DIM_KEYS = ('hidden', 'cond_dim', 'n_blocks', 'n_heads')
ALPHABET = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'
ENS = os.path.join(ROOT, 'oracle', '_ref', 'CreiLOV_ens')


def _marginals(tok, K=20):
    return torch.nn.functional.one_hot(tok.long(), K).float().mean(0)


def _entropy(p):
    return float(-(p * torch.log(p.clamp_min(1e-12))).sum(-1).mean())


def test_free_run_real_dims_fitness_oracle_statistics():
    if not os.path.isdir(ENS) or len(os.listdir(ENS)) != 10:
        pytest.skip('oracle/_ref/CreiLOV_ens missing: run __graft_entry__.build() where the reference checkout exists')
    import sgpo_amd
    sgpo_amd.install_dropin()
    from fitness.inference_oracle import inference_oracle
    from models.mdlm import MDLM
    from problem.protein import ProteinPredictor
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    from sgpo_amd import weights as W
    g = load_golden('free_run_real')
    T, temp = int(g['T']), float(g['temp'])
    Lq = g['x'].shape[1]
    full_seq = bytes(g['full_seq'].numpy().tolist()).decode()
    dims = dict(zip(DIM_KEYS, [int(v) for v in g['dims']]))
    sd = W.dit_synthetic_state(seed=int(g['seed']), vocab=31, **dims)
    cfg = dict(diffusion='absorbing_state', parameterization='subs', time_conditioning=False,
               model=dict(hidden_size=dims['hidden'], cond_dim=dims['cond_dim'], n_blocks=dims['n_blocks'],
                          n_heads=dims['n_heads'], length=1024))
    net = MDLM('synthetic', cfg, None, num_steps=T, seq_len=Lq, device='cuda:0', state_dict=sd)
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq=full_seq, residues=None)
    pred = ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0')
    pred.model.load_state_dict(W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=Lq))
    pred.eval()
    algo = Classifier_Guidance_Inpaint(net, dc, forward_op=pred, temperature=temp, device='cuda:0')
    N, n_ref = 2048, g['x'].shape[0]
    net.reseed(4321)
    np.random.seed(7)
    x, seqs = algo.inference(num_samples=N, detokenize=True)
    assert len(seqs) == N and all(len(s) == Lq and set(s) <= set(ALPHABET[:20]) for s in seqs)
    _, y = inference_oracle(seqs, protein='CreiLOV', model_path=ENS, full_seq=full_seq, device='cuda:0')
    y = y.double()
    vf = pred.log_likelihood(x).cpu().double()
    tok = torch.tensor([[ALPHABET.index(c) for c in s] for s in seqs])
    # the reference's 64
    ref_tok = g['repaired'].long()
    ref_y = torch.as_tensor(g['fitness']).double()
    ref_vf = torch.as_tensor(g['value_function']).double()
    # device fitness kernel vs the reference's own numbers on the reference's own strings (row f-1 on the real weights)
    ref_seqs = [''.join(ALPHABET[int(t)] for t in row) for row in ref_tok]
    _, y_chk = inference_oracle(ref_seqs, protein='CreiLOV', model_path=ENS, full_seq=full_seq, device='cuda:0')
    torch.testing.assert_close(y_chk.double(), ref_y, rtol=1e-5, atol=1e-6)
    pooled = _marginals(tok)

    def stats(t, yy, vv):
        return dict(tv=float(0.5 * (_marginals(t) - pooled).abs().sum(-1).mean()), entropy=_entropy(_marginals(t)),
                    fitness_mean=float(yy.mean()), fitness_std=float(yy.std()), value_function_mean=float(vv.mean()))
    ref = stats(ref_tok, ref_y, ref_vf)
    gen = torch.Generator().manual_seed(0)
    sub = []
    for _ in range(200):
        idx = torch.randperm(N, generator=gen)[:n_ref]
        sub.append(stats(tok[idx], y[idx], vf[idx]))
    report = {}
    for k in ref:
        v = torch.tensor([s[k] for s in sub])
        lo, hi = float(v.quantile(0.0025)), float(v.quantile(0.9975))
        pad = 0.2 * (hi - lo)
        report[k] = (ref[k], lo - pad, hi + pad)
    print('statistic: reference(64)  [band of 64-subsets of the device run]')
    for k, (r, lo, hi) in report.items():
        print(f'  {k}: {r:.5f}  [{lo:.5f}, {hi:.5f}]')
    print(f'  device, all {N}: fitness mean {float(y.mean()):.5f}, entropy {_entropy(pooled):.4f}, unique {len(set(seqs)) / N:.3f}')
    for k, (r, lo, hi) in report.items():
        assert lo <= r <= hi, (k, r, lo, hi)
