"""CPU: oracle restatements of the D3PM wrapper and the DAPS loop vs goldens captured from the
reference import (models/d3pm.py, sampling/daps.py).  The denoiser inside is the oracle's own
ByteNet restatement (third party upstream, parity unpinned) -- the goldens store its logits, so
the wrapper arithmetic is pinned independently of it."""
import pytest
import torch

from conftest import load_golden
from oracle import bytenet as obn
from oracle import d3pm as od3pm
from oracle import daps as odaps
from oracle import mlp as omlp
from sgpo_amd import weights as W

BN_SMALL = dict(n_tokens=31, d_embedding=8, d_model=64, n_layers=3, kernel_size=5)
BN_DEV = dict(n_tokens=31, d_embedding=8, d_model=128, n_layers=3, kernel_size=5)   # the smallest the device kernels take
S = 26


def _sched():
    Q_bar, Q = od3pm.q_random_schedule(S, 500)
    return Q, Q_bar, od3pm.sigma_from_qbar(Q_bar, S)


def test_schedule_and_sigma():
    g = load_golden('d3pm_small')
    Q, Q_bar, sigma = _sched()
    assert torch.equal(sigma, g['sigma'])            # incl. sigma[499] = +inf (fp64 product cast to fp32)
    assert torch.isinf(sigma[499]) and abs(float(sigma[1]) - 0.0040) < 1e-4 and abs(float(sigma[250]) - 0.6972) < 1e-3
    # fp32 accumulation gives the finite variant
    qb32, _ = od3pm.q_random_schedule(S, 500, accumulate='fp32')
    s32 = od3pm.sigma_from_qbar(qb32, S)
    assert torch.isfinite(s32[:499]).all()


def _tags(g, default):
    return bytes(g['tags'].numpy().tolist()).decode() if 'tags' in g else default


# *_real: BASELINE config 3 at full size (TrpB L = 389, ByteNet restatement with its real dims, 1000 MH proposals)
@pytest.mark.parametrize('name', ['d3pm_small', 'd3pm_real'])
def test_p_sample_and_q_sample(name):
    g = load_golden(name)
    Q, Q_bar, sigma = _sched()
    sd = W.bytenet_synthetic_state(seed=int(g['bn_seed']), **(BN_SMALL if name == 'd3pm_small' else {}))
    x = g['x']
    for tag in _tags(g, 'abcde'):
        t, tn = int(g[f't_{tag}'][0]), int(g[f't_{tag}'][1])
        tn = None if tn < 0 else tn
        logits = obn.bytenet_forward(sd, x, torch.tensor([t] * x.shape[0]))
        torch.testing.assert_close(logits, g[f'logits_{tag}'], rtol=1e-5, atol=1e-6)
        p = od3pm.p_sample(g[f'logits_{tag}'], x, t, tn, Q, Q_bar, sigma, S, hard=False)
        torch.testing.assert_close(p, g[f'soft_{tag}'], rtol=1e-12, atol=1e-15)
        xh = od3pm.p_sample(g[f'logits_{tag}'], x, t, tn, Q, Q_bar, sigma, S, e=g[f'e_{tag}'])
        assert torch.equal(xh, g[f'hard_{tag}'])
    xq = od3pm.q_sample(x, g['tq'], Q_bar, S, g['eq'])
    assert torch.equal(xq, g['q_sample'])


@pytest.mark.parametrize('name', ['daps_small', 'daps_real'])
def test_metropolis_hasting(name):
    g = load_golden(name)
    res1 = [int(r) for r in g['residues']]
    msd = W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=len(res1))
    pred = omlp.Predictor(msd, res1)           # reads the 1-indexed list raw (defect D3)
    residues0 = [r - 1 for r in res1]          # MH proposes on the 0-indexed list
    L = g['x0hat'].shape[1]
    for tag in ('lo', 'hi', 'inf'):
        xo = odaps.metropolis_hasting(g['x0hat'], pred.log_likelihood, torch.as_tensor(g[f'sigma_{tag}']), g['r_idx'].shape[0], 128.,
                                      1000, residues0, S, L, g['r_idx'], g['r_val'], g['r_u'])
        assert torch.equal(xo, g[f'mh_{tag}']), tag
    assert (g['mh_inf'] != g['x0hat']).any()   # some proposals were accepted
    ts = odaps.time_steps(500, 50)
    assert torch.equal(ts, g['time_steps']) and ts[0] == 499 and ts[1] == 488 and ts[-1] == 1


def test_daps_inference_tiny():
    g = load_golden('daps_small')
    Q, Q_bar, sigma = _sched()
    res1 = [int(r) for r in g['residues']]
    msd = W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=len(res1))
    pred = omlp.Predictor(msd, res1)
    residues0 = [r - 1 for r in res1]
    sd = W.bytenet_synthetic_state(seed=int(g['bn_seed']), **BN_SMALL)
    ts = g['inf_time_steps']
    xt = g['inf_start']
    B, L = xt.shape
    full = torch.tensor([W_ALPHA.index(c) for c in bytes(g['full_seq'].numpy().tolist()).decode()])
    mask = torch.zeros(L, dtype=torch.int64)
    mask[residues0] = 1

    def denoise(x, t):
        return obn.bytenet_forward(sd, x, torch.tensor([t] * x.shape[0]))

    n = len(ts)
    for i in range(n):
        x0hat = odaps.uncond_sample(denoise, xt, ts[i], 4, Q, Q_bar, sigma, S, g[f'inf_ode_e_{i}'])
        x0y = odaps.metropolis_hasting(x0hat, pred.log_likelihood, sigma[ts[i]], 25, 128., 1000, residues0, S, L,
                                       g[f'inf_mh_idx_{i}'], g[f'inf_mh_val_{i}'], g[f'inf_mh_u_{i}'])
        if i == n - 1:
            xt = x0y
            break
        xt = od3pm.q_sample(x0y, ts[i + 1].repeat(B), Q_bar, S, g[f'inf_q_e_{i}'])
    xt = xt * mask + full * (1 - mask)
    assert torch.equal(xt, g['inf_x'])


W_ALPHA = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'


ALPHABET = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'


def oracle_cg_d3pm(g, tag, stop=None):
    """The restatement's Classifier_Guidance_Inpaint loop on a D3PM net from the golden's start state and noise:
    (final projected tokens, states before each step)."""
    from conftest import golden_noise_d3pm
    from oracle import cls_guidance as ocg
    from oracle import mdlm as omdlm
    Q, Q_bar, sigma = _sched()
    sd = W.bytenet_synthetic_state(seed=int(g['bn_seed']), **BN_DEV)
    B, L = g[f'x_{tag}'].shape
    T, temp = int(g['T']), float(g[f'temp_{tag}'])
    res = [int(r) for r in g['residues']]
    pred = omlp.Predictor(W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=len(res)), res) if temp else None
    x, es = golden_noise_d3pm(g, B, L)
    states = []
    for i, t in enumerate(torch.linspace(T - 1, 1, T - 1, dtype=int)[:stop]):
        states.append(x.clone())
        logits = obn.bytenet_forward(sd, x, torch.tensor([int(t)] * B))
        q = od3pm.p_sample(logits, x, int(t), None, Q, Q_bar, sigma, S, hard=False)
        xx = x
        x, _ = ocg.cg_step(q, x, lambda qm: omdlm.get_guided_rates(pred, xx, t, qm, S, guide_temp=temp)[0], es[i])
    full = torch.tensor([ALPHABET.index(c) for c in bytes(g['full_seq'].numpy().tolist()).decode()])
    mask = torch.zeros(L, dtype=torch.int64)
    mask[[r - 1 for r in res]] = 1
    return ocg.project(x, mask, full), states


@pytest.mark.parametrize('tag', ['unguided', 'guided'])
def test_cls_guidance_inference_on_d3pm(tag):
    """Classifier_Guidance_Inpaint.inference on a D3PM net: the round-0 data path of pareto.py:48-67 (temperature 0) and
    the guided form (models/d3pm.py:109-199 inside sampling/cls_guidance.py:34-52), all 499 steps: the restatement
    visits the reference's states one by one and reaches its projected result."""
    g = load_golden('cg_d3pm_small')
    x, states = oracle_cg_d3pm(g, tag)
    want = g[f'states_{tag}'].long()
    for i, s_ in enumerate(states):
        assert torch.equal(s_, want[i]), f'state before step {i}'
    assert torch.equal(x, g[f'x_{tag}'])


def test_daps_inference_on_mdlm():
    """DAPS on an MDLM net (sampling/daps.py, self.type == "MDLM"): the restatement replays the reference's tiny full
    inference call by call from the recorded draws -- hard MDLM.p_sample (defects D1 repaired, D2 kept), the
    Metropolis-Hastings chain over 31 target states with sigma from net.sigma(t), MDLM.q_sample -- and reaches its
    result bit for bit."""
    from oracle import cls_guidance as ocg
    from oracle import mdlm as omdlm
    g = load_golden('daps_mdlm_small')
    dims = dict(zip(('hidden', 'cond_dim', 'n_blocks', 'n_heads'), [int(v) for v in g['dims']]))
    sd = W.dit_synthetic_state(seed=int(g['seed']), vocab=31, **dims)
    B, L = g['x'].shape
    T, S31 = int(g['T']), 31
    res1 = [int(r) for r in g['residues']]
    res0 = [r - 1 for r in res1]
    pred = omlp.Predictor(W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=len(res1)), res1)
    ts = odaps.time_steps(T, int(g['num_steps']))
    assert torch.equal(ts, g['time_steps'])
    n_ode = int(g['ode_steps']) - 1
    xt = torch.full((B, L), 28, dtype=torch.int64)
    ip = 0
    for i in range(int(g['num_steps'])):
        inner = torch.linspace(int(ts[i]), 0, int(g['ode_steps']), dtype=int)
        for k in range(n_ode):
            assert int(inner[k]) == int(g['p_t'][ip]) and torch.equal(xt, g['p_in'][ip])
            xt = omdlm.p_sample(sd, xt, inner[k], T, 28, hard=True, u=g[f'inf_ode_u_{i}'][k])
            assert torch.equal(xt, g['p_out'][ip]), f'p_sample call {ip}'
            ip += 1
        sigma = omdlm.total_noise(ts[i] / T)                    # MDLM.sigma (models/mdlm.py:100-102)
        torch.testing.assert_close(sigma.float(), g['mh_sigma'][i], rtol=1e-6, atol=0)
        assert torch.equal(xt, g['mh_in'][i])
        x0 = odaps.metropolis_hasting(xt, pred.log_likelihood, sigma, int(g['mh_steps']), float(g['alpha']), float(g['beta']),
                                      res0, S31, L, g[f'inf_mh_idx_{i}'], g[f'inf_mh_val_{i}'], g[f'inf_mh_u_{i}'])
        assert torch.equal(x0, g['mh_out'][i]), f'metropolis_hasting call {i}'
        if i < int(g['num_steps']) - 1:
            xt = omdlm.q_sample(x0, ts[i + 1].repeat(B), T, 28, g[f'inf_q_u_{i}'])
            assert torch.equal(xt, g['q_out'][i])
        else:
            xt = x0
    full = torch.tensor([ALPHABET.index(c) for c in bytes(g['full_seq'].numpy().tolist()).decode()])
    mask = torch.zeros(L, dtype=torch.int64)
    mask[res0] = 1
    assert torch.equal(ocg.project(xt, mask, full), g['x'])
