"""GPU parity of the whole MDLM + classifier-guidance loop through the drop-in host layer.

The oracle trajectory is pinned bit-exactly to the reference (tests/test_oracle_mdlm.py).  The
device path is then checked three ways, as `north_star` asks:
  1. teacher-forced steps: at every oracle state, same injected noise -> identical tokens wherever
     the race margin exceeds the logit tolerance (bf16 denoiser: 2e-2);
  2. per-step guided probabilities within the bf16 tolerance;
  3. free-running batches: per-position marginals and value-function mean agree statistically.
"""
import pytest
import torch

from conftest import golden_noise, load_golden
from tf_common import TeacherForced, oracle_soft_step

pytestmark = pytest.mark.gpu

DIM_KEYS = ('hidden', 'cond_dim', 'n_blocks', 'n_heads')
ALPHABET = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'


@pytest.fixture(scope='module')
def env():
    import sgpo_amd
    sgpo_amd.install_dropin()
    from sgpo_amd import runtime
    runtime.init('cuda:0')
    return runtime


def _build(g, T, Lq, full_seq, temp, mlp_seed):
    from models.mdlm import MDLM
    from problem.protein import ProteinPredictor
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    from sgpo_amd import weights as W
    dims = dict(zip(DIM_KEYS, [int(v) for v in g['dims']]))
    sd = W.dit_synthetic_state(seed=int(g['seed']), vocab=31, **dims)
    cfg = dict(diffusion='absorbing_state', parameterization='subs', time_conditioning=False,
               model=dict(hidden_size=dims['hidden'], cond_dim=dims['cond_dim'], n_blocks=dims['n_blocks'],
                          n_heads=dims['n_heads'], length=1024))
    net = MDLM('synthetic', cfg, None, num_steps=T, seq_len=Lq, device='cuda:0', state_dict=sd)
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq=full_seq, residues=None)
    pred = ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0')
    msd = W.mlp_synthetic_state(seed=mlp_seed, n_residues=Lq)
    pred.model.load_state_dict(msd)
    pred.eval()
    algo = Classifier_Guidance_Inpaint(net, dc, forward_op=pred, temperature=temp, device='cuda:0')
    return net, pred, algo, sd, msd


@pytest.mark.parametrize('tag', ['small_unguided', 'small_guided', 'real_guided'])
def test_cg_teacher_forced_against_reference_trajectory(env, tag):
    """`real_guided` is BASELINE configs 1/2 at full size: DiT 512 x 8 blocks, CreiLOV, 127 steps, temperature 0.064."""
    from oracle import cls_guidance as ocg
    from oracle import mdlm as omdlm
    from oracle import mlp as omlp
    g = load_golden(f'cg_{tag}')
    T, temp = int(g['T']), float(g['temp'])
    B, Lq = g['x'].shape
    if 'es' not in g:
        g['es'] = golden_noise(g, B, Lq)
    full_seq = bytes(g['full_seq'].numpy().tolist()).decode()
    net, pred, algo, sd, msd = _build(g, T, Lq, full_seq, temp, int(g['mlp_seed']))
    opred = omlp.Predictor(msd)
    # oracle trajectory (== the reference's, see test_oracle_mdlm.py)
    x_fin, trace = ocg.inference_mdlm(sd, B, Lq, T, 28, g['es'], predictor=opred, temperature=temp, return_trace=True)
    assert torch.equal(x_fin, g['x'])
    states = [28 * torch.ones(B, Lq, dtype=torch.int64)] + trace[:-1]
    timesteps = torch.linspace(T - 1, 1, T - 1, dtype=int)
    tf = TeacherForced()
    for i, t in enumerate(timesteps):
        xs = states[i]
        e = g['es'][i]
        q, zmax = oracle_soft_step(sd, xs, t, T)
        x_ref, p_ref = ocg.cg_step(q, xs, lambda qm: omdlm.get_guided_rates(opred, xs, t, qm, 31, temp)[0], e)
        assert torch.equal(x_ref, trace[i])
        net.injected_noise.append(e)
        xo, p = net.guided_step(xs.cuda(), t, pred, temp, want_probs=True)
        tf.step(i, p.cpu(), p_ref, xo.cpu(), x_ref, e, zmax)
    tf.finish(f'cg_{tag}')


def test_cg_inference_end_to_end_and_drop_in_api(env):
    """`Algo.inference(detokenize=True)` returns (tokens, strings) like the reference; free-running
    device marginals match the oracle's statistically (B=256 vs B=256, small model)."""
    from oracle import cls_guidance as ocg
    from oracle import mlp as omlp
    g = load_golden('cg_small_guided')
    T, temp = int(g['T']), float(g['temp'])
    Lq = g['x'].shape[1]
    full_seq = bytes(g['full_seq'].numpy().tolist()).decode()
    net, pred, algo, sd, msd = _build(g, T, Lq, full_seq, temp, int(g['mlp_seed']))
    B = 256
    net.reseed(123)
    x, seqs = algo.inference(num_samples=B, detokenize=True)
    assert x.shape == (B, Lq) and x.dtype == torch.int64 and len(seqs) == B
    assert all(len(s) == Lq and set(s) <= set(ALPHABET[:20]) for s in seqs)
    # same seed -> same samples; different seed -> different
    net.reseed(123)
    x2 = algo.inference(num_samples=B)
    assert torch.equal(x, x2)
    net.reseed(124)
    x3 = algo.inference(num_samples=B)
    assert not torch.equal(x, x3)
    # oracle free run with its own Exp(1) noise
    gen = torch.Generator().manual_seed(5)
    es = torch.empty(T - 1, B, Lq, 31).exponential_(1, generator=gen)
    xo = ocg.inference_mdlm(sd, B, Lq, T, 28, es, predictor=omlp.Predictor(msd), temperature=temp)
    oh_d = torch.nn.functional.one_hot(x.cpu(), 31).float().mean(0)
    oh_o = torch.nn.functional.one_hot(xo, 31).float().mean(0)
    tv = 0.5 * (oh_d - oh_o).abs().sum(-1)          # per-position total variation
    # two independent B=256 draws of a 31-way categorical: E[TV] ~ 0.12; bound the mean loosely
    assert float(tv.mean()) < 0.2, f'mean per-position TV {float(tv.mean()):.3f}'
    f_d = pred.log_likelihood(x).mean().item()
    f_o = omlp.Predictor(msd).log_likelihood(xo).mean().item()
    assert abs(f_d - f_o) < 0.15 * max(1.0, abs(f_o)), (f_d, f_o)


def test_free_run_statistics_4096_samples(env):
    """SURVEY.md section 4, statistical level: 4096 free-running device samples against 1024 free-running oracle
    samples (4 independent batches of 256, own noise) -- per-position marginals (total variation), per-position
    Shannon entropy (diversity) and value-function mean."""
    from oracle import cls_guidance as ocg
    from oracle import mlp as omlp
    g = load_golden('cg_small_guided')
    T, temp = int(g['T']), float(g['temp'])
    Lq = g['x'].shape[1]
    full_seq = bytes(g['full_seq'].numpy().tolist()).decode()
    net, pred, algo, sd, msd = _build(g, T, Lq, full_seq, temp, int(g['mlp_seed']))
    net.reseed(777)
    x = algo.inference(num_samples=4096).cpu()
    gen = torch.Generator().manual_seed(11)
    xo = []
    for _ in range(4):
        es = torch.empty(T - 1, 256, Lq, 31).exponential_(1, generator=gen)
        xo.append(ocg.inference_mdlm(sd, 256, Lq, T, 28, es, predictor=omlp.Predictor(msd), temperature=temp))
    xo = torch.cat(xo)
    pd_ = torch.nn.functional.one_hot(x, 31).float().mean(0)
    po = torch.nn.functional.one_hot(xo, 31).float().mean(0)
    tv = 0.5 * (pd_ - po).abs().sum(-1)
    # sampling noise alone: E[TV] of two empirical 31-way distributions with 4096 and 1024 draws is ~0.05-0.07
    assert float(tv.mean()) < 0.10 and float(tv.max()) < 0.20, (float(tv.mean()), float(tv.max()))
    ent = lambda p: -(p * torch.log(p.clamp_min(1e-12))).sum(-1)
    hd, ho = ent(pd_), ent(po)
    # the plug-in entropy of 1024 draws is biased low by ~(k-1)/(2n) ~ 0.01 nats; allow 0.06 on the mean
    assert abs(float(hd.mean() - ho.mean())) < 0.06, (float(hd.mean()), float(ho.mean()))
    with torch.no_grad():
        fd = pred.log_likelihood(x.cuda()).cpu()
        fo = omlp.Predictor(msd).log_likelihood(xo)
    se = float((fd.var() / 4096 + fo.var() / 1024).sqrt())
    assert abs(float(fd.mean() - fo.mean())) < 4 * se + 1e-3, (float(fd.mean()), float(fo.mean()), se)
    # unique fraction (the diversity number update_data sees) agrees as well
    ud = len({tuple(r.tolist()) for r in x[:1024]}) / 1024
    uo = len({tuple(r.tolist()) for r in xo}) / 1024
    assert abs(ud - uo) < 0.05, (ud, uo)


def test_get_guided_rates_signature(env):
    """The reference-signature entry point (models/mdlm.py:340-413) on the fused jump kernel."""
    g = load_golden('cg_small_guided')
    T, temp = int(g['T']), float(g['temp'])
    Lq = g['x'].shape[1]
    full_seq = bytes(g['full_seq'].numpy().tolist()).decode()
    net, pred, algo, sd, msd = _build(g, T, Lq, full_seq, temp, int(g['mlp_seed']))
    xs = g['xs'].cuda()
    q = net.p_sample(xs, torch.tensor(9), hard=False)
    torch.testing.assert_close(q.cpu(), g['q'], rtol=3e-2, atol=1e-5)
    qg = net.get_guided_rates(pred, xs, torch.tensor(9), g['q'].cuda().clone(), guide_temp=temp)
    torch.testing.assert_close(qg.cpu(), g['qg'], rtol=2e-3, atol=1e-7)
    assert net.get_guided_rates(pred, xs, 9, q, guide_temp=0) is q


def test_get_guided_rates_taylor_branch(env):
    """`use_tag=True` (models/mdlm.py:377-394) through the drop-in: against the reference's own branch run with a value
    function that takes one-hot rows (golden `tag_small`), and with the drop-in `ProteinPredictor`, whose MLPModel accepts
    one-hot rows (the reference's does not), against torch autograd on the CPU copy of the same weights.  D3PM shares
    the helper: its estimate for a predictor that is linear in the one-hot rows equals its exact jump ratios."""
    from oracle import mdlm as omdlm
    from oracle.toy_predictor import OneHotToy
    g = load_golden('tag_small')
    gs = load_golden('cg_small_guided')
    T = int(g['T'])
    xs = g['xs']
    Lq = xs.shape[1]
    full_seq = bytes(gs['full_seq'].numpy().tolist()).decode()
    net, pred, algo, sd, msd = _build(gs, T, Lq, full_seq, 0.5, int(gs['mlp_seed']))
    toy = OneHotToy(Lq, 31, seed=int(g['toy_seed'])).cuda()
    for temp, key in zip(g['temps'].tolist(), ('qg_a', 'qg_b')):
        qg = net.get_guided_rates(toy, xs.cuda(), torch.tensor(9), g['q'].cuda().clone(), use_tag=True, guide_temp=temp)
        torch.testing.assert_close(qg.cpu(), g[key], rtol=1e-4, atol=1e-9)
    # the shipped predictor class: gradient through MLPModel at the one-hot point
    qg = net.get_guided_rates(pred, xs.cuda(), torch.tensor(9), g['q'].cuda().clone(), use_tag=True, guide_temp=0.5)
    import copy
    cpu_model = copy.deepcopy(pred.model).cpu()
    ref, _ = omdlm.get_guided_rates(lambda x1h, t: cpu_model(x1h, t), xs, torch.tensor(9), g['q'].clone(), 31, guide_temp=0.5,
                                    use_tag=True)
    torch.testing.assert_close(qg.cpu(), ref.detach(), rtol=1e-4, atol=1e-9)
    assert not torch.equal(qg.cpu(), g['q'])


@pytest.mark.parametrize('name', ['mdlm_sample_small', 'mdlm_sample_real'])
def test_mdlm_sample_uncond_teacher_forced(env, name):
    """Row a-15, `MDLM.sample` (Uncond path; mdlm_diffusion.py:634-700), walked along the REFERENCE's own states (golden
    `states`: what the reference handed to its denoiser before every update and before the final argmax):
    per update the device's reverse probabilities within the logit tolerance and its tokens equal to the reference's next
    state wherever the race cannot flip inside it; carry-over exact; the noise-removal argmax equal wherever the top
    two log-probabilities are further apart than the tolerance.  `mdlm_sample_real`: real DiT dims, CreiLOV length."""
    from models.mdlm import MDLM
    from oracle import dit as odit
    from oracle import mdlm as omdlm
    from sgpo_amd import weights as W
    from tf_common import LOGIT_RTOL
    g = load_golden(name)
    dims = dict(zip(DIM_KEYS, [int(v) for v in g['dims']]))
    sd = W.dit_synthetic_state(seed=int(g['seed']), vocab=31, **dims)
    cfg = dict(model=dict(hidden_size=dims['hidden'], cond_dim=dims['cond_dim'], n_blocks=dims['n_blocks'],
                          n_heads=dims['n_heads']))
    B, Lq = g['x'].shape
    T = int(g['T'])
    states = g['states'].long()
    assert states.shape[0] == T + 1 and torch.equal(states[0], torch.full((B, Lq), 28))
    net = MDLM('synthetic', cfg, None, num_steps=T, seq_len=Lq, device='cuda:0', state_dict=sd)
    timesteps = torch.linspace(1, 1e-5, T + 1)
    dt = (1 - 1e-5) / T
    tf = TeacherForced()
    for i in range(T):
        xs, u = states[i], g['us'][i]
        z = odit.dit_forward(sd, xs, torch.zeros(B))
        logp = omdlm.subs_parameterization(z, xs, 28)
        mt = 1 - torch.exp(-omdlm.total_noise(timesteps[i]))
        ms = 1 - torch.exp(-omdlm.total_noise(timesteps[i] - dt))
        q_ref = logp.exp() * (mt - ms)
        q_ref[:, :, 28] = ms
        x_ref = omdlm.ddpm_update(sd, xs, timesteps[i] * torch.ones(B, 1), dt, 28, u)
        assert torch.equal(x_ref, states[i + 1])                  # the oracle walks the reference's trajectory
        net.injected_noise.append(u)
        xo, q = net.ddpm_step(xs.cuda(), i, want_probs=True)
        masked = xs == 28
        # the race of _sample_categorical is q / (1e-10 - log(u + 1e-10)): hand the helper that denominator
        ok = tf.step(i, q.cpu()[masked], q_ref[masked], xo.cpu()[masked], x_ref[masked], (1e-10 - (u + 1e-10).log())[masked],
                     float(z.abs().max()))
        assert torch.equal(xo.cpu()[~masked], xs[~masked])        # carry-over
    tf.finish(name)
    # noise removal: plain argmax of the subs-parameterised log-probabilities at the last state
    xs = states[T]
    z = odit.dit_forward(sd, xs, torch.zeros(B))
    logp = omdlm.subs_parameterization(z, xs, 28)
    assert torch.equal(logp.argmax(-1), g['x'])
    xo = net.noise_removal(xs.cuda()).cpu()
    top2 = logp.topk(2, dim=-1).values
    sure = (top2[..., 0] - top2[..., 1]) > 2 * LOGIT_RTOL * float(z.abs().max())
    assert torch.equal(xo[sure], g['x'][sure]) and float(sure.float().mean()) > 0.9
    # and the free run on the same uniforms stays close to the reference's result
    for u in g['us']:
        net.injected_noise.append(u)
    x = net.sample(B).cpu()
    agree = (x == g['x']).float().mean().item()
    print(f'{name}: free run agrees with the reference on {agree:.4f} of the positions')
    assert agree > 0.9, agree
    assert int((x == 28).sum()) == 0


def test_daps_on_mdlm_teacher_forced_against_reference_calls(env):
    """DAPS on an MDLM net (sampling/daps.py, self.type == "MDLM") against the reference's own tiny inference
    (daps_mdlm_small.npz, every call recorded; the restatement replays it bit for bit, tests/test_oracle_d3pm.py).  Each of
    the reference's calls is replayed on the device from the reference's inputs and draws: hard MDLM.p_sample (tokens
    wherever the race margin exceeds the bf16 logit bound, carry-over exact), the Metropolis-Hastings chain over 31
    target states (`sgpo_daps_mh`, bit-exact), MDLM.q_sample (bit-exact).  Then the whole loop through the drop-in API on
    the same draws: it must land on the reference's result wherever no near-tie was met."""
    from models.mdlm import MDLM
    from oracle import dit as odit
    from oracle import mdlm as omdlm
    from problem.protein import ProteinPredictor
    from sampling.daps import DAPS
    from sgpo_amd import weights as W
    g = load_golden('daps_mdlm_small')
    dims = dict(zip(DIM_KEYS, [int(v) for v in g['dims']]))
    sd = W.dit_synthetic_state(seed=int(g['seed']), vocab=31, **dims)
    B, Lq = g['x'].shape
    T = int(g['T'])
    res1 = [int(r) for r in g['residues']]
    full = bytes(g['full_seq'].numpy().tolist()).decode()
    cfg = dict(diffusion='absorbing_state', parameterization='subs', time_conditioning=False,
               model=dict(hidden_size=dims['hidden'], cond_dim=dims['cond_dim'], n_blocks=dims['n_blocks'],
                          n_heads=dims['n_heads'], length=1024))
    net = MDLM('synthetic', cfg, None, num_steps=T, seq_len=Lq, device='cuda:0', state_dict=sd)
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq=full, residues=res1)
    pred = ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0')
    pred.model.load_state_dict(W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=len(res1)))
    pred.eval()
    algo = DAPS(net, forward_op=pred, num_steps=int(g['num_steps']), ode_steps=int(g['ode_steps']), mh_steps=int(g['mh_steps']),
                alpha=float(g['alpha']), beta=float(g['beta']), data_config=dc, device='cuda:0')
    assert algo.type == 'MDLM' and torch.equal(algo.time_steps, g['time_steps'])
    n_ode = int(g['ode_steps']) - 1
    tf = TeacherForced(min_checked=0.7)
    for ip in range(g['p_in'].shape[0]):
        i, k = divmod(ip, n_ode)
        xs, t, u = g['p_in'][ip], g['p_t'][ip], g[f'inf_ode_u_{i}'][k]
        z = odit.dit_forward(sd, xs, torch.zeros(B))
        q_ref = omdlm.p_sample(sd, xs, t, T, 28, hard=False, log_p_x0=omdlm.subs_parameterization(z, xs, 28))
        net.injected_noise.append(u)
        xo = net.p_sample(xs.cuda(), t).cpu()
        q = net.p_sample(xs.cuda(), t, hard=False).cpu()
        masked = xs == 28
        assert torch.equal(xo[~masked], xs[~masked])                          # carry-over
        if bool(masked.any()):
            tf.step(ip, q[masked], q_ref[masked], xo[masked], g['p_out'][ip][masked], (1e-10 - (u + 1e-10).log())[masked],
                    float(z.abs().max()))
    tf.finish('daps_mdlm_small p_sample calls')
    for i in range(g['mh_in'].shape[0]):
        algo.injected_mh.append((g[f'inf_mh_idx_{i}'], g[f'inf_mh_val_{i}'], g[f'inf_mh_u_{i}']))
        x0 = algo.metropolis_hasting(g['mh_in'][i].cuda(), pred, None, net.sigma(algo.time_steps[i]), steps=algo.mh_steps,
                                     alpha=algo.alpha).cpu()
        assert torch.equal(x0, g['mh_out'][i]), f'metropolis_hasting call {i}'
        torch.testing.assert_close(net.sigma(algo.time_steps[i]).float().cpu(), g['mh_sigma'][i], rtol=1e-6, atol=0)
    assert bool((g['mh_in'] != g['mh_out']).any())                            # the chains do move
    for i in range(g['q_in'].shape[0]):
        net.injected_noise.append(g[f'inf_q_u_{i}'])
        assert torch.equal(net.q_sample(g['q_in'][i].cuda(), g['q_t'][i]).cpu(), g['q_out'][i])
    # the whole loop on the reference's draws
    for i in range(algo.num_steps):
        for k in range(n_ode):
            net.injected_noise.append(g[f'inf_ode_u_{i}'][k])
        algo.injected_mh.append((g[f'inf_mh_idx_{i}'], g[f'inf_mh_val_{i}'], g[f'inf_mh_u_{i}']))
        if i < algo.num_steps - 1:
            net.injected_noise.append(g[f'inf_q_u_{i}'])
    x = algo.inference(num_samples=B, verbose=False).cpu()
    agree = float((x == g['x']).float().mean())
    print(f'daps_mdlm_small: whole loop agrees with the reference on {agree:.4f} of the positions')
    assert agree > 0.9 and not net.injected_noise and not algo.injected_mh


def test_shard_invariance_of_full_inference(env):
    """Two emulated ranks (row offsets 0 and B/2, same key) reproduce the single-rank batch bit for bit:
    rows are independent in every kernel and the Philox counters are keyed by the global row."""
    from sgpo_amd.dist import shard_bounds
    g = load_golden('cg_small_guided')
    T, temp = int(g['T']), float(g['temp'])
    Lq = g['x'].shape[1]
    full_seq = bytes(g['full_seq'].numpy().tolist()).decode()
    net, pred, algo, sd, msd = _build(g, T, Lq, full_seq, temp, int(g['mlp_seed']))
    B = 64
    net.reseed(77, row_offset=0)
    whole = algo.inference(num_samples=B)
    parts = []
    for r in range(3):                       # ragged 3-way split
        lo, hi = shard_bounds(B, r, 3)
        net.reseed(77, row_offset=lo)
        parts.append(algo.inference(num_samples=hi - lo))
    assert torch.equal(torch.cat(parts), whole)


def test_full_size_properties_config2(env):
    """BASELINE config 2 shapes (4096 sequences x 119 residues, DiT h512 x 8 blocks, exact guidance) on a short
    horizon (8 reverse steps), through size-independent properties: same seed -> same tokens; an 8-way ragged
    split with global row offsets (what 8 ranks compute) reproduces the unsplit batch bit for bit (the shards
    tile the GEMMs differently from the whole batch); every token is inside the vocabulary;
    value-function scores through the C ABI equal the CPU restatement on a sample of rows."""
    import bench
    from oracle import mlp as omlp
    from sgpo_amd.dist import shard_bounds
    net, pred, algo, sd, msd = bench.build('cuda:0', 4096)
    net.timestep = 9
    B = 4096
    net.reseed(5, row_offset=0)
    whole = algo.inference(num_samples=B)
    net.reseed(5, row_offset=0)
    assert torch.equal(whole, algo.inference(num_samples=B))
    parts = []
    for r in range(8):
        lo, hi = shard_bounds(B - 5, r, 8)          # 4091 rows: ragged shards
        net.reseed(5, row_offset=lo)
        parts.append(algo.inference(num_samples=hi - lo))
    assert torch.equal(torch.cat(parts), whole[:B - 5])
    # (the reference's soft step keeps move_chance_s in the mask column of unmasked positions too -- SURVEY a-3 --
    #  so a short horizon leaves masks behind; project_sequences replaces them afterwards)
    assert int(whole.min()) >= 0 and int(whole.max()) < 31 and float((whole == 28).float().mean()) < 0.25
    score = pred.log_likelihood(whole).cpu()
    ref = omlp.Predictor(msd).log_likelihood(whole[:64].cpu())
    torch.testing.assert_close(score[:64], ref, rtol=1e-4, atol=1e-5)


def test_full_size_properties_config3(env):
    """BASELINE config 3 shapes (2048 sequences per GPU x 389 residues, ByteNet 16 x d1024, DAPS with 1000 MH proposals)
    on a short horizon (2 annealing steps x 3 ODE steps): determinism, 4-way ragged shard invariance, positions
    outside the 15 designable residues equal the parent, tokens inside the 26 D3PM states."""
    from models.d3pm import D3PM
    from problem.protein import ProteinPredictor
    from sampling.daps import DAPS
    from sgpo_amd import weights as W
    from sgpo_amd.dist import shard_bounds
    res = [117, 118, 119, 162, 166, 182, 183, 184, 185, 186, 227, 228, 230, 231, 301]
    Lq, B = 389, 2048
    g = torch.Generator().manual_seed(0)
    full = ''.join(ALPHABET[int(t)] for t in torch.randint(0, 20, (Lq,), generator=g))
    net = D3PM('synthetic', Lq, device='cuda:0', state_dict=W.bytenet_synthetic_state(seed=0))
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq=full, residues=res)
    pred = ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0')
    pred.model.load_state_dict(W.mlp_synthetic_state(seed=1, n_residues=len(res)))
    pred.eval()
    algo = DAPS(net, forward_op=pred, num_steps=2, ode_steps=3, mh_steps=1000, alpha=128., beta=1000, data_config=dc,
                device='cuda:0')
    net.reseed(9, row_offset=0)
    whole = algo.inference(num_samples=B, verbose=False)
    net.reseed(9, row_offset=0)
    assert torch.equal(whole, algo.inference(num_samples=B, verbose=False))
    parts = []
    for r in range(4):
        lo, hi = shard_bounds(B - 3, r, 4)
        net.reseed(9, row_offset=lo)
        parts.append(algo.inference(num_samples=hi - lo, verbose=False))
    assert torch.equal(torch.cat(parts), whole[:B - 3])
    frozen = [l for l in range(Lq) if l not in algo.residues]
    parent = torch.tensor([ALPHABET.index(c) for c in full], device=whole.device)
    assert torch.equal(whole[:, frozen], parent[frozen].expand(B, -1))
    assert int(whole.min()) >= 0 and int(whole.max()) < 26
    assert bool((whole[:, algo.residues] != parent[algo.residues]).any())


def test_full_size_properties_config4(env):
    """BASELINE config 4 shapes (200 sequences x 389 residues, DiT h512 x 8 blocks, 5 NOS steps of size 0.5) on a short
    horizon: determinism and ragged 3-way shard invariance -- the batchmean KL of the NOS objective takes the
    unsharded batch through `net.global_batch` -- and the projection onto the 15 designable residues."""
    from models.mdlm import MDLM
    from problem.protein_nos import NOSHead
    from sampling.NOS import NOS
    from sgpo_amd import weights as W
    from sgpo_amd.dist import shard_bounds
    res = [117, 118, 119, 162, 166, 182, 183, 184, 185, 186, 227, 228, 230, 231, 301]
    Lq, B = 389, 200
    g = torch.Generator().manual_seed(0)
    full = ''.join(ALPHABET[int(t)] for t in torch.randint(0, 20, (Lq,), generator=g))
    cfg = dict(diffusion='absorbing_state', parameterization='subs', time_conditioning=False,
               model=dict(hidden_size=512, cond_dim=128, n_blocks=8, n_heads=8, length=1024))
    net = MDLM('synthetic', cfg, None, num_steps=7, seq_len=Lq, device='cuda:0', state_dict=W.dit_synthetic_state(seed=0))
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq=full, residues=res)
    torch.manual_seed(1)
    head = NOSHead(512).cuda().eval()
    algo = NOS(net, dc, forward_op=head, num_nos_steps=5, nos_step_size=0.5, nos_stability_coef=0.1, device='cuda:0')
    net.reseed(3, row_offset=0)
    whole = algo.inference(num_samples=B, verbose=False)
    net.reseed(3, row_offset=0)
    assert torch.equal(whole, algo.inference(num_samples=B, verbose=False))
    net.global_batch = B
    parts = []
    for r in range(3):
        lo, hi = shard_bounds(B, r, 3)
        net.reseed(3, row_offset=lo)
        parts.append(algo.inference(num_samples=hi - lo, verbose=False))
    net.global_batch = None
    assert torch.equal(torch.cat(parts), whole)
    frozen = [l for l in range(Lq) if l not in algo.residues]
    parent = torch.tensor([ALPHABET.index(c) for c in full], device=whole.device)
    assert torch.equal(whole[:, frozen], parent[frozen].expand(B, -1))


def test_model_sampler_matrix_remaining_combinations(env):
    """The README matrix beyond the benchmarked pairs, end to end through the drop-in: UDLM (uniform diffusion, d3pm
    parameterisation, time-conditioned) under the classifier-guided sampler on a residue subset -- its fused step is
    `sgpo_guided_step_udlm`, and its PARITY is `test_cg_udlm_teacher_forced_against_reference_trajectory` below, this
    test only adds the invariants of a whole run -- and DAPS on an MDLM net (no oracle trajectory exists for that
    pair).  Each keeps determinism under a seed, legal tokens, frozen positions equal to the parent."""
    from models.mdlm import MDLM
    from problem.protein import ProteinPredictor
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    from sampling.daps import DAPS
    from sgpo_amd import weights as W
    dims = dict(hidden=128, cond_dim=32, n_blocks=2, n_heads=2)
    sd = W.dit_synthetic_state(seed=4, vocab=31, **dims)
    Lq, res1 = 30, [3, 4, 5, 10, 11, 20]
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq='M' * Lq, residues=res1)
    pred = ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0')
    pred.model.load_state_dict(W.mlp_synthetic_state(seed=2, n_residues=len(res1)))
    pred.eval()
    frozen = [l for l in range(Lq) if l not in [r - 1 for r in res1]]
    model = dict(hidden_size=128, cond_dim=32, n_blocks=2, n_heads=2, length=64)

    def check(algo, net, n=12, **kw):
        net.reseed(31)
        a = algo.inference(num_samples=n, verbose=False, **kw)
        net.reseed(31)
        b = algo.inference(num_samples=n, verbose=False, **kw)
        assert torch.equal(a, b) and a.shape == (n, Lq) and a.dtype == torch.int64
        assert int(a.min()) >= 0 and int(a.max()) < 31
        assert bool((a[:, frozen] == ALPHABET.index('M')).all())
        return a

    udlm = MDLM('synthetic', dict(diffusion='uniform', parameterization='d3pm', time_conditioning=True, model=model), None,
                num_steps=16, seq_len=Lq, device='cuda:0', state_dict=sd)
    check(Classifier_Guidance_Inpaint(udlm, dc, forward_op=pred, temperature=0.5, device='cuda:0'), udlm)
    mdlm = MDLM('synthetic', dict(diffusion='absorbing_state', parameterization='subs', time_conditioning=False, model=model),
                None, num_steps=16, seq_len=Lq, device='cuda:0', state_dict=sd)
    x = check(DAPS(mdlm, forward_op=pred, num_steps=3, ode_steps=4, mh_steps=100, alpha=4., beta=1., data_config=dc,
                   device='cuda:0'), mdlm)
    assert bool((x[:, [r - 1 for r in res1]] != ALPHABET.index('M')).any())


def test_skip_unmasked_is_bit_identical_over_a_whole_run(env):
    """`MDLM(skip_unmasked=True)` (opt-in): sequences without a masked position skip the denoiser
    forward of that step.  A whole 127-step guided run at the real DiT dims on CreiLOV (B = 48, Philox noise) and a
    stacked Thompson round (`inference_many`, 3 draws of 10) give the tokens of the full path bit for bit, and a
    sizeable share of the rows really was skipped (defect D2: about 39 % of the sequences per step at L = 119)."""
    g = load_golden('cg_real_guided')
    T, temp = int(g['T']), float(g['temp'])
    Lq = g['x'].shape[1]
    full_seq = bytes(g['full_seq'].numpy().tolist()).decode()
    net, pred, algo, sd, msd = _build(g, T, Lq, full_seq, temp, int(g['mlp_seed']))
    B = 48
    net.reseed(99)
    x_full = algo.inference(num_samples=B)
    many_full = algo.inference_many([pred, None, pred], [5, 6, 7], num_samples=10)
    net.skip_unmasked = True
    net.reseed(99)
    x_skip = algo.inference(num_samples=B)
    frac = net.skipped_rows / (net.skipped_rows + net.denoised_rows)
    many_skip = algo.inference_many([pred, None, pred], [5, 6, 7], num_samples=10)
    net.skip_unmasked = False
    assert torch.equal(x_full, x_skip)
    assert all(torch.equal(a, b) for a, b in zip(many_full, many_skip))
    print(f'skip_unmasked: {frac:.3f} of the (sequence, step) denoiser rows skipped')
    assert 0.25 < frac < 0.5, frac


def test_deferred_nan_check_names_the_offending_step(env):
    """the sampling loop reads the kernels' NaN flags every NAN_CHECK_EVERY steps; the error must still
    name the step at which the rates went bad (the reference raises at that step, models/mdlm.py:410-411).  A value
    function whose head turns NaN from step 5 on: ValueError naming step index 5, raised at the window's check."""
    from problem.protein import ProteinPredictor
    from sgpo_amd import weights as W
    g = load_golden('cg_small_guided')
    T, temp = 40, float(g['temp'])
    Lq = g['x'].shape[1]
    full_seq = bytes(g['full_seq'].numpy().tolist()).decode()
    net, pred, algo, sd, msd = _build(g, T, Lq, full_seq, temp, int(g['mlp_seed']))
    bad_sd = {k: v.clone() for k, v in msd.items()}
    bad_sd['cls_head.2.weight'] = torch.full_like(bad_sd['cls_head.2.weight'], float('nan'))   # (the bias cancels in f(jump) - f(x))
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq=full_seq, residues=None)
    bad = ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0')
    bad.model.load_state_dict(bad_sd)
    bad.eval()
    real_step, seen = net.guided_step, []

    def step(x, t, predictor, temperature, **kw):
        seen.append(kw.get('step'))
        return real_step(x, t, bad if kw.get('step', 0) >= 5 else predictor, temperature, **kw)
    net.guided_step = step
    net.reseed(3)
    with pytest.raises(ValueError, match='NaN') as err:
        algo.inference(num_samples=4)
    assert 'step index 5' in str(err.value), str(err.value)
    assert seen[-1] == algo.NAN_CHECK_EVERY - 1          # raised at the first window check, not at the end of the run
    net.guided_step = real_step
    net.reseed(3)
    algo.inference(num_samples=4)                          # the flags were cleared: a clean run passes


UDLM_CFG = dict(diffusion='uniform', parameterization='d3pm', time_conditioning=True)    # configs/model/udlm.yaml:6-9


def _build_udlm(g, T, Lq, full_seq, temp):
    from models.mdlm import MDLM
    from problem.protein import ProteinPredictor
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    from sgpo_amd import weights as W
    dims = dict(zip(DIM_KEYS, [int(v) for v in g['dims']]))
    sd = W.dit_synthetic_state(seed=int(g['seed']), vocab=31, **dims)
    cfg = dict(UDLM_CFG, model=dict(hidden_size=dims['hidden'], cond_dim=dims['cond_dim'], n_blocks=dims['n_blocks'],
                                    n_heads=dims['n_heads'], length=1024))
    net = MDLM('synthetic', cfg, None, num_steps=T, seq_len=Lq, device='cuda:0', state_dict=sd)
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq=full_seq, residues=None)
    pred = ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0')
    msd = W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=Lq)
    pred.model.load_state_dict(msd)
    pred.eval()
    algo = Classifier_Guidance_Inpaint(net, dc, forward_op=pred if temp else None, temperature=temp, device='cuda:0')
    return net, pred, algo, sd, msd


@pytest.mark.parametrize('name,tag', [('cg_udlm_small', 'unguided'), ('cg_udlm_small', 'guided'),
                                      ('cg_udlm_real', 'unguided'), ('cg_udlm_real', 'guided')])
def test_cg_udlm_teacher_forced_against_reference_trajectory(env, name, tag):
    """UDLM under Classifier_Guidance_Inpaint (scripts/run_prior_sample.sh:12: `model=udlm`; configs/model/udlm.yaml):
    at every state of the REFERENCE's own run (goldens written by tools/refgen/make_goldens.py gen_cg_udlm) the device
    step -- time-conditioned denoiser, sgpo_guidance_ratio_mlp (all jumps), sgpo_guided_step_udlm -- gives the oracle's
    probabilities within the bf16 logit bound and the reference's next tokens wherever the race margin exceeds it.
    Temperature 0 is the prior_sample.py path, > 0 the guided one; `real` = the real DiT dims on CreiLOV."""
    from oracle import cls_guidance as ocg
    from oracle import mdlm as omdlm
    from oracle import mlp as omlp
    g = load_golden(name)
    T, temp = int(g['T']), float(g[f'temp_{tag}'])
    B, Lq = g[f'x_{tag}'].shape
    es = g[f'es_{tag}'] if f'es_{tag}' in g else golden_noise(g, B, Lq)
    full_seq = bytes(g['full_seq'].numpy().tolist()).decode()
    net, pred, algo, sd, msd = _build_udlm(g, T, Lq, full_seq, temp)
    assert net.diffusion == 'uniform' and net.time_conditioning
    opred = omlp.Predictor(msd)
    states = g[f'states_{tag}'].long()
    timesteps = torch.linspace(T - 1, 1, T - 1, dtype=int)
    tf = TeacherForced()
    for i, t in enumerate(timesteps):
        xs = states[i]
        e = es[i]
        q, zmax = oracle_soft_step(sd, xs, t, T, diffusion='uniform')
        x_ref, p_ref = ocg.cg_step(q, xs, lambda qm: omdlm.get_guided_rates(opred, xs, t, qm, 31, temp)[0], e)
        want = states[i + 1] if i + 1 < T - 1 else g[f'x_{tag}']
        assert torch.equal(x_ref, want), f'oracle left the reference trajectory at step {i}'
        net.injected_noise.append(e)
        xo, p = net.guided_step(xs.cuda(), t, algo.forward_op, algo.temperature, want_probs=True)
        tf.step(i, p.cpu(), p_ref, xo.cpu(), x_ref, e, zmax)
    tf.finish(f'{name}/{tag}')


def test_cg_udlm_sampler_takes_the_uniform_step_and_q_sample(env):
    """(a) `Classifier_Guidance_Inpaint.inference` on a UDLM net is the loop of the fused UDLM step: a free run under
    Philox equals the same steps issued one by one through `guided_step`, and differs from what the absorbing-state
    kernel would give on the same logits (the round-2 defect: a UDLM net silently took that kernel).  (b) `q_sample`
    of a uniform net (q_xt's uniform branch, mdlm_diffusion.py:598-601; training/train_classifier.py:13 calls it)
    reproduces the reference's result from the reference's two draws, and under Philox replaces a fraction
    move_chance of the positions by tokens uniform over the 31 letters."""
    from sgpo_amd import _lib as L
    rt = env
    g = load_golden('cg_udlm_small')
    T = int(g['T'])
    B, Lq = g['x_guided'].shape
    full_seq = bytes(g['full_seq'].numpy().tolist()).decode()
    net, pred, algo, sd, msd = _build_udlm(g, T, Lq, full_seq, 0.005)
    net.reseed(5)
    x_run = algo.inference(num_samples=64)
    net.reseed(5)
    x = net.get_start(64)
    for t in torch.linspace(T - 1, 1, T - 1, dtype=int):
        x = net.guided_step(x, t, pred, 0.005)
    assert torch.equal(algo.project(x), x_run)
    # the absorbing-state kernel on a UDLM state: different probabilities (carry-over one-hot vs posterior)
    xs = g['states_guided'][5].long().cuda()
    mt, ms, sig = net._move_chances(torch.tensor(9))
    logits = net._denoise(xs, sig)
    _, p_u = rt.guided_step_udlm(logits, xs, mt, ms, noise_src=rt.NoiseSource(1), want_probs=True)
    _, p_a = rt.reverse_step_mdlm(logits, xs, mt, ms, 28, L.STEP_SOFT_CG, noise_src=rt.NoiseSource(1), want_probs=True)
    assert float((p_u - p_a).abs().max()) > 0.1
    # (b) q_sample, injected draws -> the reference's tokens
    net.injected_noise.append(g['uq'])
    net.injected_noise.append(g['tokq'])
    xq = net.q_sample(g['xq'].cuda(), g['tq'])
    assert torch.equal(xq.cpu(), g['xq_out'])
    # Philox: rate and uniformity
    n = 4096
    x0 = torch.full((n, Lq), 3, dtype=torch.int64, device='cuda')
    tq = torch.full((n,), 8)
    xq = net.q_sample(x0, tq).cpu()
    mc = 0.999 * 8 / T
    hist = torch.bincount(xq.flatten(), minlength=31).double()
    moved_to_other = 1.0 - float(hist[3]) / xq.numel()
    assert abs(moved_to_other - mc * 30 / 31) < 0.01, (moved_to_other, mc)
    others = torch.cat([hist[:3], hist[4:]])
    assert float(others.min()) > 0.85 * float(others.mean()) and float(others.max()) < 1.15 * float(others.mean())
    net.reseed(5)
    a = net.q_sample(x0[:8], tq[:8])
    net.reseed(5)
    assert torch.equal(a, net.q_sample(x0[:8], tq[:8]))


@pytest.mark.parametrize('read_pos', [None, [3, 4, 9, 20, 21]])
def test_guided_step_groups_entry_point_equals_one_call_per_group(env, read_pos):
    """`sgpo_guided_step_groups_mdlm` through the C ABI: 19 stacked draws (more than one launch holds: 16), ragged
    sizes, one of them unguided, each with its own value function and Philox key -- against
    `sgpo_guidance_ratio_mlp_absorbing` + `sgpo_reverse_step_mdlm` called once per draw.  Bit-exact tokens, two steps
    in a row (the second with the call offsets advanced)."""
    from sgpo_amd import _lib as L
    from sgpo_amd import weights as W
    rt = env
    Lq, K, S, mask = 23, 31, 31, 28
    pos = list(range(Lq)) if read_pos is None else read_pos
    sizes = [3, 1, 7, 2, 5, 4, 1, 6, 2, 3, 8, 1, 2, 5, 3, 4, 2, 9, 1]
    gen = torch.Generator().manual_seed(12)
    vfs = [None if i == 5 else rt.MlpValueFunction(W.mlp_synthetic_state(seed=70 + i, n_residues=len(pos)), pos)
           for i in range(len(sizes))]
    lo = [0]
    for n in sizes:
        lo.append(lo[-1] + n)
    B = lo[-1]
    slots = None
    if read_pos is not None:
        slots = torch.full((Lq,), -1, dtype=torch.int32)
        for i, p in enumerate(pos):
            slots[p] = i
        slots = slots.cuda()
    nan_a, nan_b = torch.zeros(1, dtype=torch.int32, device='cuda'), torch.zeros(1, dtype=torch.int32, device='cuda')
    srcs_a = [rt.NoiseSource(seed=900 + i, row_offset=3 * i) for i in range(len(sizes))]
    srcs_b = [rt.NoiseSource(seed=900 + i, row_offset=3 * i) for i in range(len(sizes))]
    runner = rt.GuidedStepGroups([(vfs[i], srcs_a[i], lo[i], lo[i + 1]) for i in range(len(sizes))], S, 'cuda:0')
    for step, (t, mt, ms) in enumerate([(40.0, 0.5, 0.45), (39.0, 0.45, 0.4)]):
        logits = (torch.randn(B, Lq, K, generator=gen) * 2).cuda()
        x = torch.randint(0, 28, (B, Lq), generator=gen)
        x[torch.rand(B, Lq, generator=gen) < 0.5] = mask
        x = x.cuda()
        got = runner.step(logits, x, t, mt, ms, mask, 0.3, slots, nan_a, torch.empty_like(x))
        want = torch.empty_like(x)
        for i in range(len(sizes)):
            xs = x[lo[i]:lo[i + 1]]
            lr = vfs[i].jump_log_ratios(xs, t, S, mask_index=mask) if vfs[i] is not None else None
            rt.reverse_step_mdlm(logits[lo[i]:lo[i + 1]], xs, mt, ms, mask, L.STEP_SOFT_CG, noise_src=srcs_b[i], log_ratio=lr,
                                 slot_of_pos=slots if lr is not None else None, guide_temp=0.3, nan_flag=nan_b,
                                 x_out=want[lo[i]:lo[i + 1]])
        assert torch.equal(got, want), f'step {step}: {int((got != want).sum())} tokens differ'
    assert int(nan_a.item()) == 0 and int(nan_b.item()) == 0
