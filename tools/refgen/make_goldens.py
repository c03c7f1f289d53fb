"""Generate tests/golden/*.npz by importing the reference (/root/reference) on
CPU with the stand-ins of tools/refgen/standins.py (SURVEY.md §8c recipe).

Runs in the build container only.  The goldens hold inputs, injected noise and
the REFERENCE's outputs; weights are regenerated at test time from the seeded
initialisers of sgpo_amd/weights.py (the seed and dims are stored in each file).

    python tools/refgen/make_goldens.py [--only mdlm,mlp,...]
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
REF = '/root/reference'
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import standins  # noqa: E402

standins.install()
sys.path.insert(0, REF)

from sgpo_amd import weights as W  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
CREILOV = ('MAGLRHTFVVADATLPDCPLVYASEGFYAMTGYGPDEVLGHNARFLQGEGTDPKEVQKIRDAIKKGEACSVRLLNYRKDGTPFWNLL'
           'TVTPIKTPDGRVSKFVGVQVDVTSKTEGKALA')
TRPB_RES = [117, 118, 119, 162, 166, 182, 183, 184, 185, 186, 227, 228, 230, 231, 301]


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB')


# ----------------------------------------------------------------------------
def dit_config(hidden=512, cond_dim=128, n_blocks=8, n_heads=8, diffusion='absorbing_state',
               parameterization='subs', time_conditioning=False):
    return standins.to_attr(dict(
        diffusion=diffusion, backbone='dit', parameterization=parameterization,
        time_conditioning=time_conditioning, T=0, subs_masking=False,
        training=dict(ema=0, antithetic_sampling=True, importance_sampling=False,
                      sampling_eps=1e-3, change_of_variables=False),
        optim=dict(lr=3e-4), sampling=dict(predictor='ddpm', steps=128, noise_removal=True),
        model=dict(hidden_size=hidden, cond_dim=cond_dim, length=1024, n_blocks=n_blocks,
                   n_heads=n_heads, scale_by_sigma=True, dropout=0.1),
        noise=dict(type='loglinear'), eval_batch_size=2))


def build_ref_mdlm(dims, seed, L, T=128, **cfgkw):
    """Reference MDLM wrapper object with synthetic weights (bypasses checkpoint loading)."""
    from models.pretraining.model.mdlm_diffusion import MDLMDiffusion, _sample_categorical
    import models.mdlm as ref_mdlm
    ref_mdlm._sample_categorical = _sample_categorical  # defect D1
    cfg = dit_config(hidden=dims['hidden'], cond_dim=dims['cond_dim'], n_blocks=dims['n_blocks'],
                     n_heads=dims['n_heads'], **cfgkw)
    core = MDLMDiffusion(cfg, tokenizer={'_target_': 'evodiff.utils.Tokenizer'})
    sd = W.dit_synthetic_state(seed=seed, vocab=31, hidden=dims['hidden'], cond_dim=dims['cond_dim'],
                               n_blocks=dims['n_blocks'], n_heads=dims['n_heads'])
    missing = core.backbone.load_state_dict(sd, strict=True)
    core.eval()
    core.set_length = lambda n: None
    net = object.__new__(ref_mdlm.MDLM)
    net.tokenizer = core.tokenizer
    net.model = core
    net.S = core.vocab_size
    net.timestep = T
    net.length = L
    net.device = 'cpu'
    net.diffusion = cfg.diffusion
    net.parameterization = cfg.parameterization
    return net, core, sd


SMALL = dict(hidden=128, cond_dim=32, n_blocks=2, n_heads=2)
REAL = dict(hidden=512, cond_dim=128, n_blocks=8, n_heads=8)


def gen_dit():
    for tag, dims, B, L in (('small', SMALL, 3, 37), ('real', REAL, 2, 119)):
        net, core, sd = build_ref_mdlm(dims, seed=11, L=L)
        g = torch.Generator().manual_seed(5)
        x = torch.randint(0, 31, (B, L), generator=g)
        x[:, ::3] = 28
        for sname, sigma in (('s0', torch.zeros(B)), ('s1', torch.tensor([0.3, 1.7, 4.0][:B]))):
            with torch.no_grad():
                logits, hs = core.backbone(x, sigma, return_hidden_states=True)
                logits2 = core.backbone(x, sigma, x_emb=hs[-1] * 0.5 + 0.1)
            save(f'dit_{tag}_{sname}', seed=11, dims=np.array(list(dims.values())), x=x, sigma=sigma,
                 logits=logits, h_last=hs[-1], h1=hs[1], logits_xemb=logits2)


def gen_mdlm():
    B, L, T = 3, 37, 128
    net, core, sd = build_ref_mdlm(SMALL, seed=12, L=L, T=T)
    g = torch.Generator().manual_seed(6)
    x = torch.randint(0, 26, (B, L), generator=g)
    x[:, ::2] = 28
    out = dict(seed=12, dims=np.array(list(SMALL.values())), x=x, T=T)
    with torch.no_grad():
        for t in (127, 64, 1):
            out[f'soft_t{t}'] = net.p_sample(x, torch.tensor(t), hard=False)
            torch.manual_seed(100 + t)
            u = torch.rand(B, L, 31)
            torch.manual_seed(100 + t)
            out[f'hard_t{t}'] = net.p_sample(x, torch.tensor(t), hard=True)
            out[f'u_t{t}'] = u
        out['logp'] = core.forward(x, torch.zeros(B))
        # q_sample
        tq = torch.tensor([5, 64, 127])
        torch.manual_seed(7)
        uq = torch.rand(B, L)
        torch.manual_seed(7)
        out['q_sample'] = net.q_sample(x.clone().clamp(max=25), tq)
        out['tq'] = tq
        out['uq'] = uq
        out['x_clean'] = x.clone().clamp(max=25)
        out['sigma_64'] = net.sigma(torch.tensor(64))
    save('mdlm_small', **out)

    # UDLM (uniform diffusion, d3pm parameterization, time conditioned)
    net, core, sd = build_ref_mdlm(SMALL, seed=13, L=L, T=T, diffusion='uniform',
                                   parameterization='d3pm', time_conditioning=True)
    x = torch.randint(0, 31, (B, L), generator=g)
    out = dict(seed=13, dims=np.array(list(SMALL.values())), x=x, T=T)
    with torch.no_grad():
        for t in (127, 3):
            out[f'soft_t{t}'] = net.p_sample(x, torch.tensor(t), hard=False)
            torch.manual_seed(200 + t)
            u = torch.rand(B, L, 31)
            torch.manual_seed(200 + t)
            out[f'hard_t{t}'] = net.p_sample(x, torch.tensor(t), hard=True)
            out[f'u_t{t}'] = u
    save('udlm_small', **out)

    gen_mdlm_sample()


def gen_mdlm_sample():
    """MDLM.sample (the Uncond path, sampling/uncond.py:63-64): MDLMDiffusion._sample with the ddpm predictor and noise
    removal (mdlm_diffusion.py:661-700).  Stores the uniforms, the result and the state handed to the denoiser before
    every update and before the final argmax (spy on core.forward)."""
    for name, dims, seed, B, L, T in (('mdlm_sample_small', SMALL, 14, 3, 37, 6), ('mdlm_sample_real', REAL, 21, 2, 119, 8)):
        net, core, sd = build_ref_mdlm(dims, seed=seed, L=L, T=T)
        core.length = L
        core.config.eval_batch_size = B
        states = []
        fwd = core.forward

        def spy(x, sigma, _fwd=fwd, _states=states):
            _states.append(x.clone())
            return _fwd(x, sigma)
        core.forward = spy
        torch.manual_seed(9)
        us = [torch.rand(B, L, 31) for _ in range(T)]
        torch.manual_seed(9)
        with torch.no_grad():
            xs = net.sample(B)
        assert len(states) == T + 1
        save(name, seed=seed, dims=np.array(list(dims.values())), us=torch.stack(us), x=xs, T=T,
             states=torch.stack(states).to(torch.int8))


def data_cfg(name):
    if name == 'CreiLOV':
        return standins.to_attr(dict(alphabet=standins.ALPHABET, alphabet_size=31, seq_len=119,
                                     full_seq=CREILOV, residues=None))
    raise KeyError(name)


def build_ref_mlp(n_res_cfg, residues, seed, full_len):
    import problem.protein as ref_pp
    dc = standins.to_attr(dict(alphabet_size=31, residues=residues, full_seq='A' * full_len))
    mc = standins.to_attr(dict(hidden_dim=64))
    m = ref_pp.MLPModel(dc, mc)
    sd = W.mlp_synthetic_state(seed=seed, n_residues=n_res_cfg)
    m.load_state_dict(sd, strict=True)
    m.eval()
    pred = object.__new__(ref_pp.ProteinPredictor)
    torch.nn.Module.__init__(pred)
    pred.device = 'cpu'
    pred.model = m
    return pred, sd


def gen_mlp():
    g = torch.Generator().manual_seed(21)
    # CreiLOV-like: residues None, D = 119
    pred, sd = build_ref_mlp(119, None, seed=31, full_len=119)
    x = torch.randint(0, 31, (5, 119), generator=g)
    t = torch.tensor([0., 1., 17., 64., 127.])
    with torch.no_grad():
        y = pred(x, t)
        ll = pred.log_likelihood(x)
    save('mlp_creilov', seed=31, x=x, t=t, y=y, ll=ll)
    # TrpB-like: 15 residues, 1-indexed list used raw (defect D3)
    pred, sd = build_ref_mlp(15, TRPB_RES, seed=32, full_len=389)
    x = torch.randint(0, 26, (5, 389), generator=g)
    with torch.no_grad():
        y = pred(x, t)
        ll = pred.log_likelihood(x)
    save('mlp_trpb', seed=32, x=x, t=t, y=y, ll=ll, residues=np.array(TRPB_RES))


def gen_cg():
    """Full Classifier_Guidance_Inpaint.inference on a small MDLM, guided + unguided."""
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    B, L, T = 2, 24, 16
    net, core, sd = build_ref_mdlm(SMALL, seed=15, L=L, T=T)
    full = CREILOV[:L]
    dc = standins.to_attr(dict(alphabet=standins.ALPHABET, alphabet_size=31, seq_len=L,
                               full_seq=full, residues=None))
    pred, msd = build_ref_mlp(L, None, seed=33, full_len=L)
    for tag, temp in (('unguided', 0.0), ('guided', 0.5)):
        algo = Classifier_Guidance_Inpaint(net, dc, forward_op=pred, temperature=temp, device='cpu')
        torch.manual_seed(77)
        es = [torch.empty(B * L, 31).exponential_(1) for _ in range(T - 1)]
        torch.manual_seed(77)
        with torch.no_grad():
            x = algo.inference(num_samples=B)
        # per-step guided rates at one fixed state for logits-level parity
        g = torch.Generator().manual_seed(3)
        xs = torch.randint(0, 26, (B, L), generator=g)
        xs[:, ::4] = 28
        with torch.no_grad():
            q = net.p_sample(xs, torch.tensor(9), hard=False)
            qg = net.get_guided_rates(pred, xs, torch.tensor(9), q.clone(), guide_temp=temp)
        save(f'cg_small_{tag}', seed=15, mlp_seed=33, dims=np.array(list(SMALL.values())), T=T,
             temp=temp, es=torch.stack(es).view(T - 1, B, L, 31), x=x, xs=xs, q=q, qg=qg,
             full_seq=np.frombuffer(full.encode(), dtype=np.uint8))


def gen_tag():
    """get_guided_rates(use_tag=True) of the reference MDLM wrapper with a value function that takes one-hot rows
    (oracle/toy_predictor.py; the shipped MLPModel cannot run this branch)."""
    from oracle.toy_predictor import OneHotToy
    B, L, T = 3, 24, 16
    net, core, sd = build_ref_mdlm(SMALL, seed=15, L=L, T=T)
    toy = OneHotToy(L, 31, seed=41)
    g = torch.Generator().manual_seed(5)
    xs = torch.randint(0, 26, (B, L), generator=g)
    xs[:, ::4] = 28
    with torch.no_grad():
        q = net.p_sample(xs, torch.tensor(9), hard=False)
    out = {}
    for temp in (0.5, 0.064):
        qg = net.get_guided_rates(toy, xs, torch.tensor(9), q.clone(), use_tag=True, guide_temp=temp)
        out[f'qg_{temp}'] = qg.detach()
    save('tag_small', seed=15, toy_seed=41, dims=np.array(list(SMALL.values())), T=T, xs=xs, q=q, temps=np.array([0.5, 0.064]),
         qg_a=out['qg_0.5'], qg_b=out['qg_0.064'])


def gen_cg_real():
    """BASELINE configs 1/2 at full size: the real DiT dims, CreiLOV (L = 119), 127 reverse steps, temperature 0.064,
    B = 2.  The Exp(1) noise of torch.multinomial (3.7 MB) is not stored: the test redraws it from the same CPU
    generator seed and checks the stored checksum."""
    import time
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    B, L, T, temp, noise_seed = 2, 119, 128, 0.064, 77
    net, core, sd = build_ref_mdlm(REAL, seed=16, L=L, T=T)
    dc = standins.to_attr(dict(alphabet=standins.ALPHABET, alphabet_size=31, seq_len=L, full_seq=CREILOV, residues=None))
    pred, msd = build_ref_mlp(L, None, seed=34, full_len=L)
    algo = Classifier_Guidance_Inpaint(net, dc, forward_op=pred, temperature=temp, device='cpu')
    gen = torch.Generator().manual_seed(noise_seed)
    es = torch.stack([torch.empty(B * L, 31).exponential_(1, generator=gen) for _ in range(T - 1)])
    torch.manual_seed(noise_seed)
    t0 = time.time()
    with torch.no_grad():
        x = algo.inference(num_samples=B)
    print(f'reference inference: {time.time() - t0:.1f} s')
    with torch.no_grad():
        fit = pred.log_likelihood(x)
    save('cg_real_guided', seed=16, mlp_seed=34, dims=np.array(list(REAL.values())), T=T, temp=temp,
         noise_seed=noise_seed, es_sum=es.double().sum(), es_first=es[0, :4], x=x, fitness=fit,
         full_seq=np.frombuffer(CREILOV.encode(), dtype=np.uint8))


def _record_states(net):
    """Make the reference loop tell which state it hands to net.p_sample at every step (the reference's
    inference returns the final tokens only)."""
    states = []
    orig = net.p_sample

    def spy(x, t, *a, **k):
        states.append(x.clone())
        return orig(x, t, *a, **k)
    net.p_sample = spy
    return states


def _yaml_value(path, key):
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)[key]


def gen_cg_trpb():
    """The residue-subset form of the guided step, as every TrpB run of MDLM + cls_guidance uses it: real DiT dims,
    TrpB (L = 389, full_seq and the 15 residues of configs/data/TrpB.yaml; the value function reads the raw
    1-indexed list, defect D3), temperature 0.064, B = 2, a short horizon (T = 16).  Stores the state before every
    step and the projected result; the Exp(1) noise is redrawn from its seed by the test."""
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    B, L, T, temp, noise_seed = 2, 389, 16, 0.064, 78
    full = _yaml_value(os.path.join(REF, 'configs', 'data', 'TrpB.yaml'), 'full_seq')
    res = _yaml_value(os.path.join(REF, 'configs', 'data', 'TrpB.yaml'), 'residues')
    assert res == TRPB_RES and len(full) == L
    net, core, sd = build_ref_mdlm(REAL, seed=19, L=L, T=T)
    dc = standins.to_attr(dict(alphabet=standins.ALPHABET, alphabet_size=31, seq_len=L, full_seq=full, residues=res))
    pred, msd = build_ref_mlp(len(res), res, seed=37, full_len=L)
    algo = Classifier_Guidance_Inpaint(net, dc, forward_op=pred, temperature=temp, device='cpu')
    gen = torch.Generator().manual_seed(noise_seed)
    es = torch.stack([torch.empty(B * L, 31).exponential_(1, generator=gen) for _ in range(T - 1)])
    states = _record_states(net)
    torch.manual_seed(noise_seed)
    with torch.no_grad():
        x = algo.inference(num_samples=B)
    assert len(states) == T - 1
    save('cg_real_guided_trpb', seed=19, mlp_seed=37, dims=np.array(list(REAL.values())), T=T, temp=temp,
         noise_seed=noise_seed, es_isum=es.view(torch.int32).long().sum(), es_first=es[0, :4], x=x,
         states=torch.stack(states).to(torch.int8), residues=np.array(res),
         full_seq=np.frombuffer(full.encode(), dtype=np.uint8))


def gen_cg_cfg1():
    """BASELINE config 1: the sampler prior_sample.py:73 builds (Classifier_Guidance_Inpaint with no forward_op, i.e.
    temperature 0) on CreiLOV at the real DiT dims, B = 64, all 127 steps.  Stores the final tokens of all 64 rows and,
    for the first four rows, the state before every step."""
    import time
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    B, L, T, noise_seed, keep = 64, 119, 128, 79, 4
    net, core, sd = build_ref_mdlm(REAL, seed=20, L=L, T=T)
    dc = standins.to_attr(dict(alphabet=standins.ALPHABET, alphabet_size=31, seq_len=L, full_seq=CREILOV, residues=None))
    algo = Classifier_Guidance_Inpaint(temperature=0., n_max_mutations=None, net=net, data_config=dc, device='cpu')
    assert algo.temperature == 0. and algo.forward_op is None
    gen = torch.Generator().manual_seed(noise_seed)
    es = torch.stack([torch.empty(B * L, 31).exponential_(1, generator=gen) for _ in range(T - 1)])
    states = _record_states(net)
    torch.manual_seed(noise_seed)
    t0 = time.time()
    with torch.no_grad():
        x = algo.inference(num_samples=B)
    print(f'reference inference (config 1, B = {B}): {time.time() - t0:.1f} s')
    save('cg_real_unguided_b64', seed=20, dims=np.array(list(REAL.values())), T=T, temp=0.0, noise_seed=noise_seed,
         es_isum=es.view(torch.int32).long().sum(), es_first=es[0, :4], x=x, states_first=torch.stack(states)[:, :keep].to(torch.int8),
         full_seq=np.frombuffer(CREILOV.encode(), dtype=np.uint8))


UDLM_CFG = dict(diffusion='uniform', parameterization='d3pm', time_conditioning=True)   # configs/model/udlm.yaml:6-9


def gen_cg_udlm():
    """UDLM under the samplers (scripts/run_prior_sample.sh:12, configs/model/udlm.yaml): the reference's own
    Classifier_Guidance_Inpaint.inference on a `diffusion: uniform`, `parameterization: d3pm`, time-conditioned net --
    models/mdlm.py:87-95 + mdlm_diffusion.py:392-419 inside sampling/cls_guidance.py:34-52 -- at temperature 0 (the
    prior_sample.py:69-76 path) and > 0, with the state before every step.  `cg_udlm_small`: small DiT, noise stored;
    `cg_udlm_real`: the real DiT dims, CreiLOV, a short horizon (T = 16), noise redrawn from its seed by the test.
    Also MDLM.q_sample on the uniform net (q_xt's uniform branch, mdlm_diffusion.py:598-601) with its two draws."""
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    B, L, T = 2, 24, 16
    full = CREILOV[:L]
    dc = standins.to_attr(dict(alphabet=standins.ALPHABET, alphabet_size=31, seq_len=L, full_seq=full, residues=None))
    pred, msd = build_ref_mlp(L, None, seed=43, full_len=L)
    out = dict(seed=23, mlp_seed=43, dims=np.array(list(SMALL.values())), T=T,
               full_seq=np.frombuffer(full.encode(), dtype=np.uint8))
    for tag, temp in (('unguided', 0.0), ('guided', 0.005)):   # synthetic value functions are flat: 0.005 gives ratios of e^+-1
        net, core, sd = build_ref_mdlm(SMALL, seed=23, L=L, T=T, **UDLM_CFG)
        algo = Classifier_Guidance_Inpaint(net, dc, forward_op=pred if temp else None, temperature=temp, device='cpu')
        assert algo.temperature == temp
        torch.manual_seed(81)
        es = [torch.empty(B * L, 31).exponential_(1) for _ in range(T - 1)]
        states = _record_states(net)
        torch.manual_seed(81)
        with torch.no_grad():
            x = algo.inference(num_samples=B)
        assert len(states) == T - 1
        out.update({f'es_{tag}': torch.stack(es).view(T - 1, B, L, 31), f'x_{tag}': x, f'temp_{tag}': temp,
                    f'states_{tag}': torch.stack(states).to(torch.int8)})
    # one guided-rate evaluation at a fixed state for value-level parity + q_sample of the uniform net
    net, core, sd = build_ref_mdlm(SMALL, seed=23, L=L, T=T, **UDLM_CFG)
    g = torch.Generator().manual_seed(4)
    xs = torch.randint(0, 31, (B, L), generator=g)
    with torch.no_grad():
        q = net.p_sample(xs, torch.tensor(9), hard=False)
        qg = net.get_guided_rates(pred, xs, torch.tensor(9), q.clone(), guide_temp=0.005)
    Bq = 5
    xq = torch.randint(0, 26, (Bq, L), generator=g)
    tq = torch.tensor([1, 5, 8, 12, 15])
    torch.manual_seed(82)
    uq = torch.rand(Bq, L)
    tokq = torch.randint(0, 31, (Bq, L))
    torch.manual_seed(82)
    with torch.no_grad():
        xq_out = net.q_sample(xq, tq)
    assert bool((xq_out != xq).any())
    save('cg_udlm_small', xs=xs, q=q, qg=qg, xq=xq, tq=tq, uq=uq, tokq=tokq, xq_out=xq_out, **out)

    B, L, T, temp, noise_seed = 2, 119, 16, 0.064, 83
    dc = standins.to_attr(dict(alphabet=standins.ALPHABET, alphabet_size=31, seq_len=L, full_seq=CREILOV, residues=None))
    pred, msd = build_ref_mlp(L, None, seed=44, full_len=L)
    out = dict(seed=24, mlp_seed=44, dims=np.array(list(REAL.values())), T=T, noise_seed=noise_seed,
               full_seq=np.frombuffer(CREILOV.encode(), dtype=np.uint8))
    for tag, tmp in (('unguided', 0.0), ('guided', temp)):
        net, core, sd = build_ref_mdlm(REAL, seed=24, L=L, T=T, **UDLM_CFG)
        algo = Classifier_Guidance_Inpaint(net, dc, forward_op=pred if tmp else None, temperature=tmp, device='cpu')
        gen = torch.Generator().manual_seed(noise_seed)
        es = torch.stack([torch.empty(B * L, 31).exponential_(1, generator=gen) for _ in range(T - 1)])
        states = _record_states(net)
        torch.manual_seed(noise_seed)
        with torch.no_grad():
            x = algo.inference(num_samples=B)
        assert len(states) == T - 1
        out.update({f'x_{tag}': x, f'temp_{tag}': tmp, f'states_{tag}': torch.stack(states).to(torch.int8)})
    save('cg_udlm_real', es_isum=es.view(torch.int32).long().sum(), es_first=es[0, :4], **out)


BN_SMALL = dict(n_tokens=31, d_embedding=8, d_model=64, n_layers=3, kernel_size=5)
BN_DEV = dict(n_tokens=31, d_embedding=8, d_model=128, n_layers=3, kernel_size=5)


def build_ref_d3pm(L, bn_seed=41, bn_cfg=None):
    """Reference D3PM wrapper (models/d3pm.py) around a stand-in denoiser: the oracle's ByteNet
    restatement with small seeded weights.  Q / Q_bar come from the restated q_random_schedule."""
    import models.d3pm as ref_d3pm
    from oracle import bytenet as obn
    sd = W.bytenet_synthetic_state(seed=bn_seed, **(BN_SMALL if bn_cfg is None else bn_cfg))
    net = object.__new__(ref_d3pm.D3PM)
    tok = standins.Tokenizer()
    Q_bar, Q = tok.q_random_schedule(timesteps=500)
    net.S = tok.K
    net.model = lambda x, t: obn.bytenet_forward(sd, x, t)
    net.tokenizer = tok
    net.scheme = 'd3pm'
    net.timestep = 500
    net.Q_bar = Q_bar[:, :net.S, :net.S]
    net.Q = Q[:, :net.S, :net.S]
    net.device = 'cpu'
    net.length = L
    net.sigma = -torch.log(((net.Q_bar[:, 0, 0] - 1 / net.S) * net.S / (net.S - 1)))  # models/d3pm.py:29
    return net, sd


def gen_d3pm():
    from oracle import bytenet as obn
    N, L, S = 3, 20, 26
    net, sd = build_ref_d3pm(L)
    g = torch.Generator().manual_seed(8)
    x = torch.randint(0, S, (N, L), generator=g)
    out = dict(bn_seed=41, x=x, sigma=net.sigma)
    with torch.no_grad():
        for tag, t, tn in (('a', 300, None), ('b', 300, 250), ('c', 499, 472), ('d', 26, 0), ('e', 1, None)):
            out[f'logits_{tag}'] = net.model(x, torch.tensor([t] * N))
            out[f'soft_{tag}'] = net.p_sample(x, t, tn, hard=False)
            torch.manual_seed(300 + t)
            e = torch.empty(N * L, S, dtype=torch.float64).exponential_(1)
            torch.manual_seed(300 + t)
            out[f'hard_{tag}'] = net.p_sample(x, t, tn, hard=True)
            out[f'e_{tag}'] = e
            out[f't_{tag}'] = np.array([t, -1 if tn is None else tn])
        tq = torch.tensor([3, 250, 499])
        torch.manual_seed(17)
        eq = torch.empty(N * L, S).exponential_(1)
        torch.manual_seed(17)
        out['q_sample'] = net.q_sample(x, tq)
        out['tq'] = tq
        out['eq'] = eq
    save('d3pm_small', **out)


def gen_cg_d3pm():
    """The reference's own Classifier_Guidance_Inpaint.inference on a D3PM net -- the round-0 data path of pareto.py:48-67
    (temperature 0, all 499 steps) and the guided form (models/d3pm.py:109-199 inside sampling/cls_guidance.py:34-52) --
    around the small ByteNet restatement, on a residue subset.  Stores the state before every step and the projected
    result; the fp64 Exp(1) draws of torch.multinomial (fp64 rates) are redrawn by the tests from the seed: the
    reference draws get_start's torch.randint first, then one (B L, S) fp64 exponential_ per step."""
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    B, L, S, noise_seed = 2, 24, 26, 85
    res = [2, 5, 6, 11, 17, 20]
    full = CREILOV[:L]
    dc = standins.to_attr(dict(alphabet=standins.ALPHABET, alphabet_size=31, seq_len=L, full_seq=full, residues=res))
    pred, msd = build_ref_mlp(len(res), res, seed=45, full_len=L)
    out = dict(bn_seed=44, mlp_seed=45, noise_seed=noise_seed, residues=np.array(res), full_seq=np.frombuffer(full.encode(), dtype=np.uint8))
    for tag, temp in (('unguided', 0.0), ('guided', 0.005)):
        net, sd = build_ref_d3pm(L, bn_seed=44, bn_cfg=BN_DEV)     # d_model 128: the smallest the device kernels take
        algo = Classifier_Guidance_Inpaint(net, dc, forward_op=pred if temp else None, temperature=temp, device='cpu')
        T = net.timestep
        torch.manual_seed(noise_seed)
        x0 = torch.randint(0, S, (B, L)).to(torch.long)
        es = torch.stack([torch.empty(B * L, S, dtype=torch.float64).exponential_(1) for _ in range(T - 1)])
        states = _record_states(net)
        torch.manual_seed(noise_seed)
        with torch.no_grad():
            x = algo.inference(num_samples=B)
        assert len(states) == T - 1 and torch.equal(states[0], x0)
        out.update({f'x_{tag}': x, f'temp_{tag}': temp, f'states_{tag}': torch.stack(states).to(torch.int8)})
    out['es_isum'] = es.view(torch.int64).sum()          # wrap-around integer sum of the bit patterns: order independent
    out['es_first'] = es[0, :4]
    save('cg_d3pm_small', T=T, **out)


def gen_d3pm_real():
    """BASELINE config 3 at full size, one call each: D3PM.p_sample / q_sample at the TrpB length around the
    ByteNet restatement with its real dims (d1024 x 16 layers; third party upstream, parity unpinned: its logits are
    stored), and DAPS.metropolis_hasting with the config's own 1000 proposals, alpha 128, beta 1000 on the 15 TrpB
    residues."""
    from sampling.daps import DAPS
    N, L, S = 2, 389, 26
    net, sd = build_ref_d3pm(L, bn_seed=43, bn_cfg={})
    g = torch.Generator().manual_seed(8)
    x = torch.randint(0, S, (N, L), generator=g)
    out = dict(bn_seed=43, x=x, sigma=net.sigma, tags=np.frombuffer(b'bc', dtype=np.uint8))
    with torch.no_grad():
        for tag, t, tn in (('b', 300, 250), ('c', 499, 472)):
            out[f'logits_{tag}'] = net.model(x, torch.tensor([t] * N))
            out[f'soft_{tag}'] = net.p_sample(x, t, tn, hard=False)
            torch.manual_seed(300 + t)
            e = torch.empty(N * L, S, dtype=torch.float64).exponential_(1)
            torch.manual_seed(300 + t)
            out[f'hard_{tag}'] = net.p_sample(x, t, tn, hard=True)
            out[f'e_{tag}'] = e
            out[f't_{tag}'] = np.array([t, -1 if tn is None else tn])
        tq = torch.tensor([3, 499])
        torch.manual_seed(17)
        eq = torch.empty(N * L, S).exponential_(1)
        torch.manual_seed(17)
        out['q_sample'] = net.q_sample(x, tq)
        out['tq'] = tq
        out['eq'] = eq
    save('d3pm_real', **out)

    N = 4
    dc = standins.to_attr(dict(alphabet=standins.ALPHABET, alphabet_size=31, seq_len=L, full_seq='A' * L, residues=TRPB_RES))
    pred, msd = build_ref_mlp(len(TRPB_RES), TRPB_RES, seed=36, full_len=L)
    algo = DAPS(net, forward_op=pred, num_steps=50, ode_steps=20, mh_steps=1000, alpha=128., max_dist=1, beta=1000,
                data_config=dc, device='cpu')
    n_res = len(algo.residues)
    x0hat = torch.randint(0, S, (N, L), generator=g)
    steps = 1000
    out = dict(mlp_seed=36, residues=np.array(TRPB_RES), x0hat=x0hat, time_steps=algo.time_steps)
    for tag, sig in (('lo', net.sigma[11]), ('hi', net.sigma[400]), ('inf', net.sigma[499])):
        torch.manual_seed(55)
        r_idx, r_val, r_u = [], [], []
        for _ in range(steps):
            r_idx.append(torch.randint(n_res, (N,)))
            r_val.append(torch.randint(S, (N,)))
            r_u.append(torch.rand(N))
        torch.manual_seed(55)
        with torch.no_grad():
            xo = algo.metropolis_hasting(x0hat, pred, None, sig, steps=steps, alpha=algo.alpha)
        out[f'mh_{tag}'] = xo
        out[f'sigma_{tag}'] = sig
    out['r_idx'], out['r_val'], out['r_u'] = torch.stack(r_idx), torch.stack(r_val), torch.stack(r_u)
    save('daps_real', **out)


def gen_daps():
    """DAPS.metropolis_hasting and a tiny DAPS.inference on the reference (sampling/daps.py)."""
    from sampling.daps import DAPS
    N, L, S = 4, 40, 26
    residues_1idx = [5, 6, 7, 20, 21, 33]
    net, bsd = build_ref_d3pm(L)
    full = CREILOV[:L]
    dc = standins.to_attr(dict(alphabet=standins.ALPHABET, alphabet_size=31, seq_len=L, full_seq=full,
                               residues=residues_1idx))
    pred, msd = build_ref_mlp(len(residues_1idx), residues_1idx, seed=35, full_len=L)
    algo = DAPS(net, forward_op=pred, num_steps=3, ode_steps=4, mh_steps=25, alpha=128., max_dist=1, beta=1000,
                data_config=dc, device='cpu')
    n_res = len(algo.residues)
    g = torch.Generator().manual_seed(9)
    x0hat = torch.randint(0, S, (N, L), generator=g)
    steps = 60
    out = dict(mlp_seed=35, bn_seed=41, residues=np.array(residues_1idx), x0hat=x0hat,
               full_seq=np.frombuffer(full.encode(), dtype=np.uint8))
    for tag, sig in (('lo', net.sigma[11]), ('hi', net.sigma[400]), ('inf', net.sigma[499])):
        torch.manual_seed(55)
        r_idx, r_val, r_u = [], [], []
        for _ in range(steps):
            r_idx.append(torch.randint(n_res, (N,)))
            r_val.append(torch.randint(S, (N,)))
            r_u.append(torch.rand(N))
        torch.manual_seed(55)
        with torch.no_grad():
            xo = algo.metropolis_hasting(x0hat, pred, None, sig, steps=steps, alpha=algo.alpha)
        out[f'mh_{tag}'] = xo
        out[f'sigma_{tag}'] = sig
    out['r_idx'], out['r_val'], out['r_u'] = torch.stack(r_idx), torch.stack(r_val), torch.stack(r_u)
    out['time_steps'] = DAPS(net, forward_op=pred, num_steps=50, ode_steps=20, data_config=dc, device='cpu').time_steps

    # tiny full inference: record every draw in call order
    B = 2
    torch.manual_seed(91)
    rec = {}
    rec['start'] = torch.randint(0, S, (B, L))
    ts = algo.time_steps
    for i in range(algo.num_steps):
        rec[f'ode_e_{i}'] = torch.stack([torch.empty(B * L, S, dtype=torch.float64).exponential_(1)
                                         for _ in range(algo.ode_steps - 1)])
        a, b, c = [], [], []
        for _ in range(algo.mh_steps):
            a.append(torch.randint(n_res, (B,)))
            b.append(torch.randint(S, (B,)))
            c.append(torch.rand(B))
        rec[f'mh_idx_{i}'], rec[f'mh_val_{i}'], rec[f'mh_u_{i}'] = torch.stack(a), torch.stack(b), torch.stack(c)
        if i < algo.num_steps - 1:
            rec[f'q_e_{i}'] = torch.empty(B * L, S).exponential_(1)
    torch.manual_seed(91)
    with torch.no_grad():
        xf = algo.inference(num_samples=B, verbose=False)
    out.update({'inf_' + k: v for k, v in rec.items()})
    out['inf_x'] = xf
    out['inf_time_steps'] = ts
    save('daps_small', **out)


def gen_daps_mdlm():
    """DAPS on an MDLM net (sampling/daps.py with self.type == "MDLM": net.sigma(t) a function, net.S = 31 target states in
    the proposals, hard MDLM.p_sample in the few-step reverse pass -- defect D1 repaired, defect D2 as is), a tiny full
    inference of the reference.  Records the inputs, the draws and the output of EVERY net.p_sample, metropolis_hasting
    and net.q_sample call in call order, so that each can be replayed on its own (teacher forcing)."""
    from sampling.daps import DAPS
    B, L, T, S = 2, 24, 16, 31
    residues_1idx = [2, 5, 6, 11, 17, 20]
    net, core, sd = build_ref_mdlm(SMALL, seed=25, L=L, T=T)
    full = CREILOV[:L]
    dc = standins.to_attr(dict(alphabet=standins.ALPHABET, alphabet_size=31, seq_len=L, full_seq=full, residues=residues_1idx))
    pred, msd = build_ref_mlp(len(residues_1idx), residues_1idx, seed=46, full_len=L)
    algo = DAPS(net, forward_op=pred, num_steps=3, ode_steps=4, mh_steps=25, alpha=4., max_dist=1, beta=1., data_config=dc,
                device='cpu')     # beta 1: proposals are accepted now and then (beta 1000 freezes the chain at these sigmas)
    assert algo.type == 'MDLM'
    n_res = len(algo.residues)
    # the draws in the order the loop makes them
    torch.manual_seed(92)
    rec = {}
    for i in range(algo.num_steps):
        rec[f'ode_u_{i}'] = torch.stack([torch.rand(B, L, S) for _ in range(algo.ode_steps - 1)])
        a, b, c = [], [], []
        for _ in range(algo.mh_steps):
            a.append(torch.randint(n_res, (B,)))
            b.append(torch.randint(S, (B,)))
            c.append(torch.rand(B))
        rec[f'mh_idx_{i}'], rec[f'mh_val_{i}'], rec[f'mh_u_{i}'] = torch.stack(a), torch.stack(b), torch.stack(c)
        if i < algo.num_steps - 1:
            rec[f'q_u_{i}'] = torch.rand(B, L)
    calls = {'p_in': [], 'p_t': [], 'p_out': [], 'mh_in': [], 'mh_sigma': [], 'mh_out': [], 'q_in': [], 'q_t': [], 'q_out': []}
    p_orig, mh_orig, q_orig = net.p_sample, algo.metropolis_hasting, net.q_sample

    def p_spy(x, t, *a_, **k_):
        o = p_orig(x, t, *a_, **k_)
        calls['p_in'].append(x.clone()); calls['p_t'].append(int(t)); calls['p_out'].append(o.clone())
        return o

    def mh_spy(x0hat, op, y, sigma, **k_):
        o = mh_orig(x0hat, op, y, sigma, **k_)
        calls['mh_in'].append(x0hat.clone()); calls['mh_sigma'].append(torch.as_tensor(sigma).reshape(())); calls['mh_out'].append(o.clone())
        return o

    def q_spy(x, t):
        o = q_orig(x, t)
        calls['q_in'].append(x.clone()); calls['q_t'].append(t.clone()); calls['q_out'].append(o.clone())
        return o
    net.p_sample, algo.metropolis_hasting, net.q_sample = p_spy, mh_spy, q_spy
    torch.manual_seed(92)
    with torch.no_grad():
        xf = algo.inference(num_samples=B, verbose=False)
    assert len(calls['p_in']) == algo.num_steps * (algo.ode_steps - 1) and len(calls['mh_in']) == algo.num_steps
    out = {'inf_' + k: v for k, v in rec.items()}
    out.update({k: torch.stack([torch.as_tensor(v) for v in vs]) for k, vs in calls.items()})
    save('daps_mdlm_small', seed=25, mlp_seed=46, dims=np.array(list(SMALL.values())), T=T, residues=np.array(residues_1idx),
         full_seq=np.frombuffer(full.encode(), dtype=np.uint8), num_steps=3, ode_steps=4, mh_steps=25, alpha=4., beta=1.,
         time_steps=algo.time_steps, x=xf, **out)


def gen_nos():
    """MDLM._nos_denoise (models/mdlm.py:184-338) with the reference DITClassifier head
    (dit.py:420-515, x_emb path: mean-pool + Linear(hidden, 1))."""
    for name, dims, B, L, seed, cases in (
            ('nos_small', SMALL, 3, 21, 16, (('a', 100, 5, 0.5, 0.1), ('b', 3, 2, 0.5, 10.0), ('c', 64, 5, 0.5, 200.0))),
            # BASELINE config 4 at full size: the real DiT dims, TrpB length, 5 NOS steps of size 0.5
            ('nos_real', REAL, 2, 389, 18, (('a', 100, 5, 0.5, 1.0), ('c', 64, 5, 0.5, 200.0)))):
        _gen_nos_case(name, dims, B, L, seed, cases)


def _gen_nos_case(name, dims, B, L, seed, cases):
    from models.pretraining.model.mdlm.models.dit import DITClassifier
    T = 128
    net, core, sd = build_ref_mdlm(dims, seed=seed, L=L, T=T)
    cm = standins.to_attr(dict(classifier_model=dict(hidden_size=dims['hidden'], cond_dim=dims['cond_dim'],
                                                     n_blocks=1, n_heads=dims['n_heads'], scale_by_sigma=True,
                                                     dropout=0.1, pooling='mean')))
    clf = DITClassifier(cm, vocab_size=31)
    g = torch.Generator().manual_seed(44)
    w_cls = torch.randn(1, dims['hidden'], generator=g) * 0.5
    b_cls = torch.randn(1, generator=g)
    with torch.no_grad():
        clf.output_layer.weight.copy_(w_cls)
        clf.output_layer.bias.copy_(b_cls)
    clf.eval()
    x = torch.randint(0, 26, (B, L), generator=g)
    x[:, ::2] = 28
    x[1, :] = 28
    out = dict(seed=seed, dims=np.array(list(dims.values())), T=T, x=x, w_cls=w_cls, b_cls=b_cls,
               tags=np.frombuffer(''.join(c[0] for c in cases).encode(), dtype=np.uint8))
    for tag, t, nsteps, lr, coef in cases:
        torch.manual_seed(500 + t)
        u = torch.rand(B, L, 31)
        torch.manual_seed(500 + t)
        xs, probs, _ = net._nos_denoise(classifier_model=clf, num_nos_steps=nsteps, nos_step_size=lr,
                                        nos_stability_coef=coef, xt=x.clone(), t=torch.tensor(t))
        out[f'xs_{tag}'] = xs
        out[f'probs_{tag}'] = probs.detach()
        out[f'u_{tag}'] = u
        out[f'hp_{tag}'] = np.array([t, nsteps, lr, coef])
    save(name, **out)


def gen_oracle_ens():
    """f-1: the reference's own inference_oracle (oracle/inference_oracle.py) on seeded synthetic
    ensembles written to a temporary model_path, plus a generation-time check of the CPU restatement
    against the reference with the checkpoints it ships (their weights are not committed)."""
    import tempfile
    import types
    bio = standins._mod('Bio')
    seqio = standins._mod('Bio.SeqIO')

    def _parse(path, fmt):
        recs, cur = [], None
        for line in open(path):
            line = line.strip()
            if line.startswith('>'):
                cur = types.SimpleNamespace(id=line[1:], seq='')
                recs.append(cur)
            elif cur is not None:
                cur.seq += line
        return recs
    seqio.parse = _parse
    seqio.read = lambda path, fmt: _parse(path, fmt)[0]
    bio.SeqIO = seqio
    standins._mod('wandb')
    cwd = os.getcwd()
    os.chdir(REF)                       # the reference opens data/<protein>/parent.fasta relative to cwd
    try:
        # the reference's `oracle` package (no __init__.py) shares its name with this repo's test
        # oracle/ directory: bind the name to the reference directory for the duration of this generator
        saved = {k: v for k, v in sys.modules.items() if k == 'oracle' or k.startswith('oracle.')}
        for k in saved:
            del sys.modules[k]
        pkg = types.ModuleType('oracle')
        pkg.__path__ = [os.path.join(REF, 'oracle')]
        sys.modules['oracle'] = pkg
        from oracle.inference_oracle import inference_oracle
        import importlib.util
        spec = importlib.util.spec_from_file_location('oracle_ens_port', os.path.join(ROOT, 'oracle', 'oracle_ensemble.py'))
        port = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(port)
        g = torch.Generator().manual_seed(77)
        aa = port.AA20
        out = {}
        for protein, n_members, seed in (('CreiLOV', 10, 11), ('TrpB', 20, 12)):
            parent = str(_parse(f'data/{protein}/parent.fasta', 'fasta')[0].seq)
            if protein == 'CreiLOV':
                n_res, base = len(parent), parent
            else:
                n_res, base = len(TRPB_RES), ''.join(parent[r - 1] for r in TRPB_RES)
            # mutants of the parent (combo) at 0 .. n_res mutated positions, so the penalty branch is taken
            seqs = []
            for n_mut in [0, 1, 5, 30, 69, 70, 71, 72, 90, 110, 119][:11 if protein == 'CreiLOV' else 4] + [3, 8, 15][: 0 if protein == 'CreiLOV' else 3]:
                n_mut = min(n_mut, n_res)
                s = list(base)
                for pos in torch.randperm(n_res, generator=g)[:n_mut].tolist():
                    s[pos] = aa[(aa.index(s[pos]) + 1 + int(torch.randint(19, (1,), generator=g))) % 20]
                seqs.append(''.join(s))
            for _ in range(5):
                seqs.append(''.join(aa[i] for i in torch.randint(20, (n_res,), generator=g).tolist()))
            states = W.oracle_ensemble_synthetic_state(seed=seed, n_members=n_members, n_res=n_res)
            with tempfile.TemporaryDirectory() as d:
                for m, sd in enumerate(states):
                    torch.save(sd, os.path.join(d, f'model{m}.pth'))
                _, y = inference_oracle(seqs, protein=protein, model_path=d)
                _, y_np = inference_oracle(seqs, protein=protein, model_path=d, impose_penalty=False)
            mine = port.ensemble_fitness(states, seqs, parent, protein)
            assert torch.allclose(mine, y, rtol=1e-6, atol=1e-7), (mine - y).abs().max()   # fp32 BLAS summation order
            print('  max |restatement - reference| synthetic', float((mine - y).abs().max()))
            # the shipped checkpoints: restatement vs reference, checked here only
            real = [torch.load(f'oracle/checkpoints/{protein}/{f}', weights_only=True, map_location='cpu')
                    for f in sorted(os.listdir(f'oracle/checkpoints/{protein}'))]
            # the shipped tensors were saved from CUDA; the reference's torch.load has no map_location, so
            # give this CPU-only process a default one (safe loader: weights_only stays at torch's default True)
            _orig_load = torch.load
            torch.load = lambda f, **kw: _orig_load(f, **{'map_location': 'cpu', 'weights_only': True, **kw})
            try:
                _, y_real = inference_oracle(seqs, protein=protein, model_path=f'oracle/checkpoints/{protein}')
            finally:
                torch.load = _orig_load
            mine_real = port.ensemble_fitness(real, seqs, parent, protein)
            assert torch.allclose(mine_real, y_real, rtol=1e-6, atol=1e-7), (mine_real - y_real).abs().max()
            print(protein, 'reference == restatement on synthetic and shipped ensembles; y range', float(y.min()), float(y.max()),
                  'real', float(y_real.min()), float(y_real.max()))
            out[f'{protein}_seqs'] = np.array(seqs)
            out[f'{protein}_parent'] = np.array(parent)
            out[f'{protein}_y'] = y.numpy()
            out[f'{protein}_y_nopenalty'] = y_np.numpy()
            out[f'{protein}_seed'] = seed
            out[f'{protein}_members'] = n_members
        save('oracle_ens', **out)
    finally:
        os.chdir(cwd)
        for k in [k for k in sys.modules if k == 'oracle' or k.startswith('oracle.')]:
            del sys.modules[k]
        sys.modules.update(saved)


def gen_free_run():
    """north_star's third check at real dims, joined to row f-1: a guided FREE run of the reference (MDLM +
    Classifier_Guidance_Inpaint, real DiT dims, CreiLOV, all 127 steps, temperature 0.064, detokenize=True ->
    project_sequences) of 64 sequences, scored by the reference's own inference_oracle on the SHIPPED CreiLOV ensemble
    (oracle/checkpoints/CreiLOV, weights-only load).  Stores the tokens, the repaired strings and their fitness; the GPU
    test compares per-position marginals, Shannon entropy and fitness mean of >= 1024 device samples against these."""
    import time
    import types
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    B, L, T, temp, seed = 64, 119, 128, 0.064, 91
    net, core, sd = build_ref_mdlm(REAL, seed=16, L=L, T=T)
    dc = standins.to_attr(dict(alphabet=standins.ALPHABET, alphabet_size=31, seq_len=L, full_seq=CREILOV, residues=None))
    pred, msd = build_ref_mlp(L, None, seed=34, full_len=L)
    algo = Classifier_Guidance_Inpaint(net, dc, forward_op=pred, temperature=temp, device='cpu')
    torch.manual_seed(seed)
    np.random.seed(seed)
    t0 = time.time()
    with torch.no_grad():
        x, seqs = algo.inference(num_samples=B, detokenize=True)
    print(f'reference free run, B = {B}: {time.time() - t0:.1f} s')
    with torch.no_grad():
        vf = pred.log_likelihood(x)
    # the reference's fitness oracle on its shipped ensemble (same module juggling as gen_oracle_ens)
    bio = standins._mod('Bio')
    seqio = standins._mod('Bio.SeqIO')

    def _parse(path, fmt):
        recs, cur = [], None
        for line in open(path):
            line = line.strip()
            if line.startswith('>'):
                cur = types.SimpleNamespace(id=line[1:], seq='')
                recs.append(cur)
            elif cur is not None:
                cur.seq += line
        return recs
    seqio.parse = _parse
    seqio.read = lambda path, fmt: _parse(path, fmt)[0]
    bio.SeqIO = seqio
    standins._mod('wandb')
    cwd = os.getcwd()
    os.chdir(REF)
    saved = {k: v for k, v in sys.modules.items() if k == 'oracle' or k.startswith('oracle.')}
    _orig_load = torch.load
    try:
        for k in saved:
            del sys.modules[k]
        pkg = types.ModuleType('oracle')
        pkg.__path__ = [os.path.join(REF, 'oracle')]
        sys.modules['oracle'] = pkg
        from oracle.inference_oracle import inference_oracle
        parent = str(_parse('data/CreiLOV/parent.fasta', 'fasta')[0].seq)
        assert parent == CREILOV
        torch.load = lambda f, **kw: _orig_load(f, **{'map_location': 'cpu', 'weights_only': True, **kw})
        _, y = inference_oracle(seqs, protein='CreiLOV', model_path='oracle/checkpoints/CreiLOV')
        _, y_np = inference_oracle(seqs, protein='CreiLOV', model_path='oracle/checkpoints/CreiLOV', impose_penalty=False)
        files = os.listdir('oracle/checkpoints/CreiLOV')
    finally:
        torch.load = _orig_load
        os.chdir(cwd)
        for k in [k for k in sys.modules if k == 'oracle' or k.startswith('oracle.')]:
            del sys.modules[k]
        sys.modules.update(saved)
    aa = standins.ALPHABET
    toks = torch.tensor([[aa.index(c) for c in s_] for s_ in seqs])
    save('free_run_real', seed=16, mlp_seed=34, dims=np.array(list(REAL.values())), T=T, temp=temp, run_seed=seed, x=x.to(torch.int8),
         repaired=toks.to(torch.int8), fitness=y.numpy(), fitness_nopenalty=y_np.numpy(), value_function=vf,
         member_files=np.array(files), full_seq=np.frombuffer(CREILOV.encode(), dtype=np.uint8))
    print('fitness mean', float(y.mean()), 'no penalty', float(y_np.mean()), 'value function mean', float(vf.mean()))


def gen_acquire():
    """f-2: the reference's own Algo.project_sequences (sampling/base.py:40-63) and
    ProteinPredictorDataset.update_data (dataset/protein.py:153-213) on scripted inputs.  numpy's
    random.choice is replaced, for the duration of the projection calls, by a reader of the noise arrays the
    restatement and the kernel take (u_special, keys); everything else is the reference's code."""
    import json
    import types
    from sampling.base import Algo
    import importlib.util
    spec = importlib.util.spec_from_file_location('acq_port', os.path.join(ROOT, 'oracle', 'acquisition.py'))
    port = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(port)

    class _A(Algo):
        def inference(self, observation=None, num_samples=1, **kw):
            raise NotImplementedError

    rng = np.random.default_rng(5)
    tok = standins.Tokenizer()
    alpha = standins.ALPHABET
    out = {}
    for case, (L, N, n_max, p_special, p_mut) in dict(a=(24, 48, None, 0.08, 0.3), b=(24, 48, 3, 0.08, 0.3),
                                                     c=(119, 32, 10, 0.01, 0.15), d=(119, 32, 200, 0.0, 0.5)).items():
        full = rng.integers(0, 20, L)
        x = np.tile(full, (N, 1))
        mut = rng.random((N, L)) < p_mut
        x[mut] = rng.integers(0, 20, int(mut.sum()))
        spec_m = rng.random((N, L)) < p_special
        x[spec_m] = rng.integers(20, 31, int(spec_m.sum()))
        x[0] = full                                                    # an untouched parent
        u_special = rng.random((N, 11)).astype(np.float32)
        keys = rng.random((N, L)).astype(np.float32)
        keys[1, : L // 2] = 0.5                                        # ties: the smaller position wins
        full_s = ''.join(alpha[t] for t in full)
        dc = standins.to_attr(dict(alphabet=alpha, alphabet_size=31, seq_len=L, full_seq=full_s, residues=None))
        algo = _A(net=types.SimpleNamespace(tokenizer=tok), data_config=dc, n_max_mutations=n_max, device='cpu')
        seqs = [''.join(alpha[t] for t in row) for row in x]
        # queue of step-1 draws: one per special character met while walking each original string
        q1 = []
        for n in range(N):
            for j in range(L):
                if x[n, j] >= 20:
                    q1.append(alpha[min(int(np.float32(u_special[n, x[n, j] - 20]) * np.float32(20)), 19)])
        state = dict(i1=0, n2=0)

        def choice(a, size=None, replace=True, p=None):
            if size is None:                                           # base.py:47
                assert list(a) == list(alpha[:20])
                v = q1[state['i1']]
                state['i1'] += 1
                return v
            assert replace is False                                    # base.py:55
            n = state['n2']
            state['n2'] += 1
            return np.array(sorted(list(a), key=lambda j: (float(keys[n, j]), j))[:size], dtype=np.int64)
        orig = np.random.choice
        np.random.choice = choice
        try:
            res = algo.project_sequences(seqs)
        finally:
            np.random.choice = orig
        assert state['i1'] == len(q1) and state['n2'] == (N if n_max is not None else 0)
        y = np.array([[tok.a_to_i[c] for c in r] for r in res])
        mine = port.project_tokens(x, full, 20, u_special, n_max, keys)
        assert np.array_equal(mine, y), case
        out.update({f'{case}_x': x, f'{case}_full': full, f'{case}_u_special': u_special, f'{case}_keys': keys,
                    f'{case}_n_max': -1 if n_max is None else n_max, f'{case}_y': y})
        print('  project case', case, 'changed rows', int((y != x).any(axis=1).sum()), 'restatement == reference')
    save('acquire_project', **out)

    # ---- update_data: the reference's class, built without its __init__ (no csv / fasta / oracle files) ----
    saved = {k: v for k, v in sys.modules.items() if k == 'oracle' or k.startswith('oracle.')}
    for k in saved:
        del sys.modules[k]
    try:
        pkg = types.ModuleType('oracle')
        pkg.__path__ = [os.path.join(REF, 'oracle')]
        sys.modules['oracle'] = pkg
        bio = standins._mod('Bio')
        bio.SeqIO = standins._mod('Bio.SeqIO')
        standins._mod('wandb')
        import pandas as pd
        import dataset.protein as dp
        dp.inference_oracle = lambda combos, protein=None, model_path=None: (None, torch.zeros(len(combos)))

        def new_ds(n_res, L, df=None):
            ds = object.__new__(dp.ProteinPredictorDataset)
            ds.seqs = ds.round_seqs = ds.y = ds.summary_df = None
            ds.round = 0
            ds.residues = list(range(L)) if n_res == L else [1, 3, 4, 6]
            ds.n_residues = n_res
            ds.df = df
            ds.data_config = types.SimpleNamespace(name='CreiLOV', oracle_model_path='unused')
            return ds

        def pool(n, L, n_distinct):
            base = [''.join(alpha[t] for t in rng.integers(0, 20, L)) for _ in range(n_distinct)]
            return [base[i] for i in rng.integers(0, n_distinct, n)]

        scenarios = {}
        L = 12
        scripts = dict(
            round_scope=[dict(n=40, distinct=25, n_needed=100, BO=False, round=0, unique_only=True),
                         dict(n=40, distinct=25, n_needed=100, BO=False, round=0, unique_only=True),
                         dict(n=40, distinct=30, n_needed=5, BO=False, round=0, unique_only=True),
                         dict(n=40, distinct=25, n_needed=100, BO=False, round=1, unique_only=True),
                         dict(n=40, distinct=25, n_needed=100, BO=False, round=1, unique_only=True),
                         dict(n=0, distinct=1, n_needed=10, BO=False, round=1, unique_only=True)],
            bo=[dict(n=30, distinct=20, n_needed=100, BO=True, round=0, unique_only=True),
                dict(n=30, distinct=20, n_needed=100, BO=True, round=0, unique_only=True),
                dict(n=30, distinct=20, n_needed=100, BO=True, round=1, unique_only=True),
                dict(n=30, distinct=28, n_needed=4, BO=True, round=2, unique_only=True),
                dict(n=30, distinct=28, n_needed=100, BO=True, round=2, unique_only=True)],
            keep_duplicates=[dict(n=20, distinct=8, n_needed=100, BO=False, round=0, unique_only=False),
                             dict(n=20, distinct=8, n_needed=100, BO=False, round=0, unique_only=False),
                             dict(n=20, distinct=8, n_needed=7, BO=False, round=1, unique_only=False)],
            empty_first=[dict(n=0, distinct=1, n_needed=10, BO=True, round=0, unique_only=True),
                         dict(n=16, distinct=6, n_needed=100, BO=True, round=0, unique_only=True)])
        shared = pool(200, L, 60)                                      # calls draw from one pool: overlaps across calls
        for name, script in scripts.items():
            ds = new_ds(L, L)
            mine = port.SequenceStore()
            calls = []
            for c in script:
                batch = [shared[i] for i in rng.integers(0, c['distinct'] * 2, c['n'])]
                before = 0 if ds.seqs is None else len(ds.seqs)
                ret = ds.update_data(list(batch), c['n_needed'], BO=c['BO'], round=c['round'], unique_only=c['unique_only'])
                added = list(ds.seqs[before:])
                assert ret == len(added)
                idx, adm = mine.update(batch, c['n_needed'], BO=c['BO'], round=c['round'], unique_only=c['unique_only'])
                assert len(idx) == len(added), (name, len(idx), len(added))
                if adm <= c['n_needed']:
                    assert sorted(batch[i] for i in idx) == sorted(added), name
                calls.append(dict(batch=batch, n_needed=c['n_needed'], BO=c['BO'], round=c['round'],
                                  unique_only=c['unique_only'], ref_added_sorted=sorted(added), ref_returned=ret,
                                  truncated=bool(adm > c['n_needed'])))
                if adm > c['n_needed']:
                    # keep both stores identical from here on: adopt the reference's choice
                    mine.seqs = list(ds.seqs)
                    mine.round_seqs = list(ds.round_seqs)
                    calls[-1]['ref_added_in_order'] = added
            scenarios[name] = calls
            print('  update scenario', name, [c['ref_returned'] for c in calls], 'restatement == reference (as sets)')
        # landscape branch (n_residues == 4): fitness looked up in the data frame, 0 when missing
        combos = [''.join(alpha[t] for t in rng.integers(0, 20, 4)) for _ in range(50)]
        df = pd.DataFrame({'Combo': combos[:40], 'fitness': rng.random(40).astype(np.float32)})
        df = df.drop_duplicates('Combo')
        ds = new_ds(4, 8, df)
        parent = ''.join(alpha[t] for t in rng.integers(0, 20, 8))
        seqs = []
        for cmb in combos:
            sq = list(parent)
            for k, r in enumerate(ds.residues):
                sq[r] = cmb[k]
            seqs.append(''.join(sq))
        import io
        import contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            ds.update_data(seqs, 1000)
        scenarios['landscape'] = dict(seqs=seqs, residues=ds.residues, combos=list(df['Combo']),
                                      fitness=[float(v) for v in df['fitness']], y=[float(v) for v in ds.y])
        with open(os.path.join(OUT, 'acquire_update.json'), 'w') as f:
            json.dump(scenarios, f)
        print('acquire_update.json: %.1f KiB' % (os.path.getsize(os.path.join(OUT, 'acquire_update.json')) / 1024))
    finally:
        for k in [k for k in sys.modules if k == 'oracle' or k.startswith('oracle.')]:
            del sys.modules[k]
        sys.modules.update(saved)


def gen_train():
    """f-3: the reference's own train_classifier (training/train_classifier.py:21-64) on scripted batches.  The
    diffusion model handed to it is a stub whose q_sample returns the pre-drawn noised batch of the step; the
    timesteps it draws with torch.randint are reproduced by re-seeding torch before the call."""
    import types
    standins._mod('hydra')
    hu = standins._mod('hydra.utils')
    hu.instantiate = lambda *a, **k: None
    sys.modules['hydra'].utils = hu
    from training.train_classifier import train_classifier
    import importlib.util
    spec = importlib.util.spec_from_file_location('oracle.train_mlp', os.path.join(ROOT, 'oracle', 'train_mlp.py'),
                                                  submodule_search_locations=None)
    import oracle.mlp  # noqa: F401  (this repo's test oracle package)
    port = importlib.util.module_from_spec(spec)
    port.__package__ = 'oracle'
    spec.loader.exec_module(port)
    out = {}
    for case, (L, residues, tc, n, bs, epochs, T) in dict(full=(24, None, True, 150, 64, 2, 128),
                                                         sites=(40, [3, 7, 8, 20, 33], True, 70, 32, 3, 128),
                                                         clean=(24, None, False, 100, 64, 2, 128)).items():
        g = torch.Generator().manual_seed(31 + L)
        n_res = L if residues is None else len(residues)
        clf, _ = build_ref_mlp(n_res, residues, 17, L)   # MLPModel keeps its time embedder either way (t = 0 when tc is off)
        x = torch.randint(0, 20, (n, L), generator=g)
        y = torch.randn(n, generator=g)
        steps = []
        for ep in range(epochs):
            perm = torch.randperm(n, generator=g)
            for i in range(0, n, bs):
                idx = perm[i:i + bs]
                xb = x[idx]
                xt = torch.where(torch.rand(xb.shape, generator=g) < 0.3, torch.full_like(xb, 28), xb) if tc else xb
                steps.append((xb, xt, y[idx]))

        class _Model:
            device = 'cpu'
            timestep = T

            def __init__(self):
                self.i = 0

            def q_sample(self, xb, t):
                xt = steps[self.i][1]
                assert torch.equal(xb, steps[self.i][0])
                self.i += 1
                return xt
        n_batches = -(-n // bs)
        loaders = [[(s[0], s[2]) for s in steps[ep * n_batches:(ep + 1) * n_batches]] for ep in range(epochs)]

        class _Loader:
            def __init__(self):
                self.ep = 0

            def __len__(self):
                return n_batches

            def __iter__(self):
                it = iter(loaders[self.ep])
                self.ep += 1
                return it
        clf.device = 'cpu'
        sd0 = {k: v.clone() for k, v in clf.model.state_dict().items()}
        cfgt = types.SimpleNamespace(learning_rate=1e-3, wandb=False, n_epochs=epochs, n_epochs_no_time=epochs)
        torch.manual_seed(99)
        ts = [torch.randint(0, T, (s[0].shape[0],)).to(torch.long) if tc else torch.zeros(s[0].shape[0], dtype=torch.long)
              for s in steps]
        torch.manual_seed(99)
        trained = train_classifier(clf, _Model(), _Loader(), cfgt, save_dir=None, project_fn=None, time_conditioned=tc)
        sd1 = {k: v.detach().clone() for k, v in trained.model.state_dict().items()}
        mine, losses = port.train(sd0, [(s[1], t, s[2]) for s, t in zip(steps, ts)], residues=residues)
        err = max(float((mine[k] - sd1[k]).abs().max()) for k in sd1)
        assert err < 1e-6, (case, err)
        print('  train case', case, len(steps), 'steps, restatement vs reference max |dw|', err, 'loss', losses[0], '->', losses[-1])
        bmax = max(s[0].shape[0] for s in steps)
        pad = lambda v, fill: torch.stack([torch.cat([q, torch.full((bmax - q.shape[0],) + q.shape[1:], fill, dtype=q.dtype)]) for q in v])
        out.update({f'{case}_xt': pad([s[1] for s in steps], 0), f'{case}_t': pad(ts, 0), f'{case}_y': pad([s[2] for s in steps], 0.),
                    f'{case}_nb': torch.tensor([s[0].shape[0] for s in steps]), f'{case}_losses': torch.tensor(losses),
                    f'{case}_L': L, f'{case}_residues': torch.tensor(residues if residues is not None else []),
                    f'{case}_tc': int(tc), f'{case}_seed': 17, f'{case}_n_res': n_res})
        for k in sd1:
            out[f'{case}_w1_{k}'] = sd1[k]
    save('train_mlp', **out)


GENS = dict(daps_mdlm=gen_daps_mdlm, cg_d3pm=gen_cg_d3pm, free_run=gen_free_run, cg_udlm=gen_cg_udlm, tag=gen_tag, train=gen_train, acquire=gen_acquire, oracle_ens=gen_oracle_ens, dit=gen_dit, mdlm=gen_mdlm, mdlm_sample=gen_mdlm_sample, mlp=gen_mlp, cg=gen_cg, cg_real=gen_cg_real, cg_trpb=gen_cg_trpb, cg_cfg1=gen_cg_cfg1, d3pm=gen_d3pm, d3pm_real=gen_d3pm_real, daps=gen_daps, nos=gen_nos)

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    names = [n for n in a.only.split(',') if n] or list(GENS)
    for n in names:
        GENS[n]()
