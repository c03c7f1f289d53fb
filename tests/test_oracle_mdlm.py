"""CPU: the oracle restatement vs goldens captured from the reference import
(tools/refgen/make_goldens.py).  fp32 vs fp32 on the same CPU -> tight tolerances."""
import numpy as np
import pytest
import torch

from conftest import golden_noise, load_golden
from oracle import cls_guidance as ocg
from oracle import dit as odit
from oracle import mdlm as omdlm
from oracle import mlp as omlp
from sgpo_amd import weights as W

DIM_KEYS = ('hidden', 'cond_dim', 'n_blocks', 'n_heads')


def _sd(g):
    dims = dict(zip(DIM_KEYS, [int(v) for v in g['dims']]))
    return W.dit_synthetic_state(seed=int(g['seed']), vocab=31, **dims)


@pytest.mark.parametrize('name', ['dit_small_s0', 'dit_small_s1', 'dit_real_s0', 'dit_real_s1'])
def test_dit_forward(name):
    g = load_golden(name)
    sd = _sd(g)
    logits, hs = odit.dit_forward(sd, g['x'], g['sigma'], return_hidden_states=True)
    torch.testing.assert_close(hs[1], g['h1'], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(hs[-1], g['h_last'], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(logits, g['logits'], rtol=1e-4, atol=2e-5)
    l2 = odit.dit_forward(sd, g['x'], g['sigma'], x_emb=g['h_last'] * 0.5 + 0.1)
    torch.testing.assert_close(l2, g['logits_xemb'], rtol=1e-4, atol=2e-5)


def test_mdlm_p_sample_and_q_sample():
    g = load_golden('mdlm_small')
    sd = _sd(g)
    T = int(g['T'])
    x = g['x']
    logp = omdlm.forward(sd, x, torch.zeros(x.shape[0]), 28)
    torch.testing.assert_close(logp, g['logp'], rtol=1e-4, atol=1e-4)
    for t in (127, 64, 1):
        q = omdlm.p_sample(sd, x, t, T, 28, hard=False)
        torch.testing.assert_close(q, g[f'soft_t{t}'], rtol=1e-4, atol=1e-7)
        xh = omdlm.p_sample(sd, x, t, T, 28, hard=True, u=g[f'u_t{t}'])
        assert torch.equal(xh, g[f'hard_t{t}'])
    # defect D2: mask column equals (1-eps) t / T^2
    q = omdlm.p_sample(sd, x, 127, T, 28, hard=False)
    assert abs(float(q[0, 0, 28]) - 0.999 * 127 / 128 / 128) < 1e-7
    xq = omdlm.q_sample(g['x_clean'], g['tq'], T, 28, g['uq'])
    assert torch.equal(xq, g['q_sample'])
    assert abs(float(omdlm.total_noise(torch.tensor(64 / T))) - float(g['sigma_64'])) < 1e-7


def test_udlm_p_sample():
    g = load_golden('udlm_small')
    sd = _sd(g)
    T = int(g['T'])
    for t in (127, 3):
        kw = dict(diffusion='uniform', parameterization='d3pm', time_conditioning=True)
        q = omdlm.p_sample(sd, g['x'], t, T, 28, hard=False, **kw)
        torch.testing.assert_close(q, g[f'soft_t{t}'], rtol=2e-4, atol=1e-7)
        xh = omdlm.p_sample(sd, g['x'], t, T, 28, hard=True, u=g[f'u_t{t}'], **kw)
        assert torch.equal(xh, g[f'hard_t{t}'])


@pytest.mark.parametrize('name', ['mdlm_sample_small', 'mdlm_sample_real'])
def test_mdlm_sample(name):
    g = load_golden(name)
    sd = _sd(g)
    B, L = g['x'].shape
    T = int(g['T'])
    x = omdlm.sample(sd, B, L, T, 28, g['us'])
    assert torch.equal(x, g['x'])
    # update by update along the reference's own states (what it handed to its denoiser)
    states = g['states'].long()
    ts, dt = torch.linspace(1, 1e-5, T + 1), (1 - 1e-5) / T
    for i in range(T):
        assert torch.equal(omdlm.ddpm_update(sd, states[i], ts[i] * torch.ones(B, 1), dt, 28, g['us'][i]), states[i + 1]), i


def test_mlp():
    g = load_golden('mlp_creilov')
    sd = W.mlp_synthetic_state(seed=int(g['seed']), n_residues=119)
    y = omlp.mlp_forward(sd, g['x'], g['t'])
    torch.testing.assert_close(y, g['y'], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(omlp.Predictor(sd).log_likelihood(g['x']), g['ll'], rtol=1e-5, atol=1e-6)
    g = load_golden('mlp_trpb')
    sd = W.mlp_synthetic_state(seed=int(g['seed']), n_residues=15)
    res = [int(r) for r in g['residues']]
    y = omlp.mlp_forward(sd, g['x'], g['t'], residues=res)
    torch.testing.assert_close(y, g['y'], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('tag', ['unguided', 'guided'])
def test_cls_guidance_inference(tag):
    g = load_golden(f'cg_small_{tag}')
    sd = _sd(g)
    temp = float(g['temp'])
    B, L = g['x'].shape
    msd = W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=L)
    pred = omlp.Predictor(msd)
    # per-step guided rates at a fixed state
    q = omdlm.p_sample(sd, g['xs'], 9, int(g['T']), 28, hard=False)
    torch.testing.assert_close(q, g['q'], rtol=1e-4, atol=1e-7)
    qg, _ = omdlm.get_guided_rates(pred, g['xs'], 9, q.clone(), 31, guide_temp=temp)
    torch.testing.assert_close(qg, g['qg'], rtol=2e-4, atol=1e-7)
    x = ocg.inference_mdlm(sd, B, L, int(g['T']), 28, g['es'], predictor=pred, temperature=temp)
    assert torch.equal(x, g['x'])


def test_cls_guidance_inference_full_size():
    """BASELINE configs 1/2 at full size (DiT 512 x 8 blocks, CreiLOV L = 119, 127 steps, temperature 0.064, B = 2):
    the restatement reaches the reference's final tokens bit for bit and its fitness values."""
    g = load_golden('cg_real_guided')
    sd = _sd(g)
    B, L = g['x'].shape
    pred = omlp.Predictor(W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=L))
    x = ocg.inference_mdlm(sd, B, L, int(g['T']), 28, golden_noise(g, B, L), predictor=pred, temperature=float(g['temp']))
    assert torch.equal(x, g['x'])
    torch.testing.assert_close(pred.log_likelihood(x), g['fitness'], rtol=1e-5, atol=1e-6)


ALPHABET = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'


def test_cls_guidance_inference_trpb_residue_subset():
    """The residue-subset form every TrpB run uses (real DiT dims, L = 389, the 15 residues of configs/data/TrpB.yaml
    read raw by the value function -- defect D3 --, temperature 0.064, B = 2, T = 16): the restatement visits the
    reference's states one by one and its projected result equals the reference's."""
    g = load_golden('cg_real_guided_trpb')
    sd = _sd(g)
    B, L = g['x'].shape
    T = int(g['T'])
    res = [int(r) for r in g['residues']]
    pred = omlp.Predictor(W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=len(res)), res)
    x, trace = ocg.inference_mdlm(sd, B, L, T, 28, golden_noise(g, B, L), predictor=pred, temperature=float(g['temp']),
                                  return_trace=True)
    states = g['states'].long()
    assert torch.equal(states[0], torch.full((B, L), 28))
    for i in range(1, T - 1):
        assert torch.equal(trace[i - 1], states[i]), f'state before step {i}'
    full = torch.tensor([ALPHABET.index(c) for c in bytes(g['full_seq'].numpy().tolist()).decode()])
    mask = torch.zeros(L, dtype=torch.int64)
    mask[[r - 1 for r in res]] = 1                       # sampling/base.py:30: the mutated set is the 0-indexed list
    assert torch.equal(ocg.project(x, mask, full), g['x'])
    assert bool((x != g['x']).any())                     # the projection is not a no-op on this run


def test_cls_guidance_config1_first_rows_teacher_forced():
    """BASELINE config 1 (prior_sample.py:73: no forward_op, temperature 0; CreiLOV, real DiT dims, B = 64, 127 steps):
    on the first four rows the restatement reproduces the reference's next state from each of its 127 states."""
    g = load_golden('cg_real_unguided_b64')
    sd = _sd(g)
    B, L = g['x'].shape
    T = int(g['T'])
    states = g['states_first'].long()
    n = states.shape[1]
    es = golden_noise(g, B, L)[:, :n]
    timesteps = torch.linspace(T - 1, 1, T - 1, dtype=int)
    for i, t in enumerate(timesteps):
        q = omdlm.p_sample(sd, states[i], t, T, 28, hard=False)
        x_next, _ = ocg.cg_step(q, states[i], lambda qm: qm, es[i])
        want = states[i + 1] if i + 1 < T - 1 else g['x'][:n]
        assert torch.equal(x_next, want), f'step {i}'


def test_taylor_guided_rates_against_reference():
    """The TAG branch of get_guided_rates (models/mdlm.py:377-394): golden = the reference's own branch run with a value
    function that takes one-hot rows (oracle/toy_predictor.py); the restatement must reproduce it."""
    from oracle.toy_predictor import OneHotToy
    g = load_golden('tag_small')
    xs, q = g['xs'], g['q']
    toy = OneHotToy(xs.shape[1], 31, seed=int(g['toy_seed']))
    for temp, key in zip(g['temps'].tolist(), ('qg_a', 'qg_b')):
        qg, _ = omdlm.get_guided_rates(toy, xs, torch.tensor(9), q.clone(), 31, guide_temp=temp, use_tag=True)
        torch.testing.assert_close(qg.detach(), g[key], rtol=1e-5, atol=1e-9)
    # for a value function that is LINEAR in the one-hot rows the first-order estimate is the exact jump ratio
    w = torch.randn(xs.shape[1] * 31, generator=torch.Generator().manual_seed(2)) * 0.1
    lin_onehot = lambda x1h, t: (x1h.flatten(1) @ w)[:, None]
    lin_tokens = lambda x, t: lin_onehot(torch.nn.functional.one_hot(x.long(), 31).float(), t)
    a, _ = omdlm.get_guided_rates(lin_onehot, xs, torch.tensor(9), q.clone(), 31, guide_temp=0.7, use_tag=True)
    b, _ = omdlm.get_guided_rates(lin_tokens, xs, torch.tensor(9), q.clone(), 31, guide_temp=0.7)
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-8)


UDLM_KW = dict(diffusion='uniform', parameterization='d3pm', time_conditioning=True)   # configs/model/udlm.yaml:6-9


@pytest.mark.parametrize('name,tag', [('cg_udlm_small', 'unguided'), ('cg_udlm_small', 'guided'),
                                      ('cg_udlm_real', 'unguided'), ('cg_udlm_real', 'guided')])
def test_cls_guidance_inference_udlm(name, tag):
    """UDLM under Classifier_Guidance_Inpaint.inference (scripts/run_prior_sample.sh:12; models/mdlm.py:87-95 +
    mdlm_diffusion.py:392-419 inside sampling/cls_guidance.py:34-52): the restatement visits every state of the
    reference's own run and reaches its final tokens, at temperature 0 (prior_sample.py) and > 0; small dims and the
    real DiT dims on CreiLOV."""
    g = load_golden(name)
    sd = _sd(g)
    T, temp = int(g['T']), float(g[f'temp_{tag}'])
    B, L = g[f'x_{tag}'].shape
    es = g[f'es_{tag}'] if f'es_{tag}' in g else golden_noise(g, B, L)
    pred = omlp.Predictor(W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=L)) if temp else None
    x, trace = ocg.inference_mdlm(sd, B, L, T, 28, es, predictor=pred, temperature=temp, return_trace=True, **UDLM_KW)
    states = g[f'states_{tag}'].long()
    assert torch.equal(states[0], torch.full((B, L), 28))       # models/mdlm.py:52-53: all-mask start, also for UDLM
    for i in range(1, T - 1):
        assert torch.equal(trace[i - 1], states[i]), f'state before step {i}'
    full = torch.tensor([ALPHABET.index(c) for c in bytes(g['full_seq'].numpy().tolist()).decode()])
    assert torch.equal(ocg.project(x, torch.ones(L, dtype=torch.int64), full), g[f'x_{tag}'])
    # a UDLM position keeps moving after it has left the mask (no carry-over): the absorbing-state step would not
    moved = sum(int(((states[i] != 28) & (states[i + 1] != states[i])).sum()) for i in range(1, T - 2))
    assert moved > 0


def test_udlm_guided_rates_and_q_sample():
    g = load_golden('cg_udlm_small')
    sd = _sd(g)
    T = int(g['T'])
    L = g['xs'].shape[1]
    pred = omlp.Predictor(W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=L))
    q = omdlm.p_sample(sd, g['xs'], 9, T, 28, hard=False, **UDLM_KW)
    torch.testing.assert_close(q, g['q'], rtol=2e-4, atol=1e-7)
    qg, _ = omdlm.get_guided_rates(pred, g['xs'], 9, q.clone(), 31, guide_temp=0.005)
    torch.testing.assert_close(qg, g['qg'], rtol=2e-4, atol=1e-7)
    assert float((g['qg'] / g['q']).max()) > 1.5      # the guidance is not a no-op at this temperature
    xq = omdlm.q_sample(g['xq'], g['tq'], T, 28, g['uq'], diffusion='uniform', uniform_tokens=g['tokq'])
    assert torch.equal(xq, g['xq_out'])
