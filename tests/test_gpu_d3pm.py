"""GPU parity of the D3PM + DAPS path: K x K posterior kernel, q_sample, ByteNet denoiser,
Metropolis-Hastings kernel, and the drop-in DAPS / Uncond loops.

The wrapper arithmetic is checked against goldens captured from the reference import
(models/d3pm.py, sampling/daps.py).  ByteNetLMTime is third party and absent: the device denoiser
is compared with the oracle's restatement only ("parity unpinned").
"""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
S = 26
ALPHABET = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'
# north_star's bf16 bound (1e-2) on denoiser outputs, relative to the range of the CPU restatement's values (ByteNet is
# "parity unpinned": the restatement is the only oracle).  Observed: 3.6e-3 ... 8.6e-3.
BN_TOL = 1e-2
BN_DEV_SMALL = dict(n_tokens=31, d_embedding=8, d_model=128, n_layers=3, kernel_size=5)


@pytest.fixture(scope='module')
def rt():
    import sgpo_amd
    sgpo_amd.install_dropin()
    from sgpo_amd import runtime
    runtime.init('cuda:0')
    return runtime


def _sched():
    from oracle import d3pm as od3pm
    Q_bar, Q = od3pm.q_random_schedule(S, 500)
    return Q, Q_bar, od3pm.sigma_from_qbar(Q_bar, S)


def _step_mats(Q, Q_bar, sigma, t, tn):
    if tn is None:
        return Q[t], Q_bar[t - 1]
    ds = sigma[tn] - sigma[t]
    return torch.eye(S) * torch.exp(ds) + torch.ones(S) / S * (1 - torch.exp(ds)), Q_bar[tn]


def _tags(g, default):
    return bytes(g['tags'].numpy().tolist()).decode() if 'tags' in g else default


# *_real: BASELINE config 3 at full size (TrpB L = 389, real ByteNet dims behind the stored logits, 1000 MH proposals)
@pytest.mark.parametrize('name', ['d3pm_small', 'd3pm_real'])
def test_posterior_kernel_against_reference(rt, name):
    g = load_golden(name)
    Q, Q_bar, sigma = _sched()
    x = g['x']
    for tag in _tags(g, 'abcde'):
        t, tn = int(g[f't_{tag}'][0]), int(g[f't_{tag}'][1])
        tn = None if tn < 0 else tn
        dq, qb = _step_mats(Q, Q_bar, sigma, t, tn)
        logits = g[f'logits_{tag}'].cuda()
        p = rt.reverse_step_d3pm(logits, x.cuda(), dq, qb, S, hard=False).cpu()
        assert p.dtype == torch.float64
        torch.testing.assert_close(p, g[f'soft_{tag}'], rtol=1e-5, atol=1e-12)   # fp32 softmax ulp -> 1e-7 relative
        e = g[f'e_{tag}'].view(x.shape[0], x.shape[1], S)
        xh = rt.reverse_step_d3pm(logits, x.cuda(), dq, qb, S, hard=True, noise=e).cpu()
        v = g[f'soft_{tag}'] / e
        top2 = v.topk(2, dim=-1).values
        ok = (top2[..., 0] - top2[..., 1]) > 1e-4 * top2[..., 0]
        assert ok.float().mean() > 0.97
        assert torch.equal(xh[ok], g[f'hard_{tag}'][ok])


@pytest.mark.parametrize('N,Lq,t,tn', [(1, 1, 300, None), (3, 7, 499, 472), (2, 389, 26, 0), (1, 600, 1, None), (5, 33, 250, 200)])
def test_posterior_shape_sweep(rt, N, Lq, t, tn):
    """One position, lengths off every tile size, both branches (Q step, analytic t_next step) and the infinite sigma:
    fp64 posterior and hard tokens against the restatement on random logits and injected Exp(1) noise."""
    from oracle import d3pm as od3pm
    Q, Q_bar, sigma = _sched()
    g = torch.Generator().manual_seed(N * 1000 + Lq)
    logits = torch.randn(N, Lq, 31, generator=g) * 2
    x = torch.randint(0, S, (N, Lq), generator=g)
    e = torch.empty(N * Lq, S, dtype=torch.float64).exponential_(1, generator=g)
    p_ref = od3pm.p_sample(logits, x, t, tn, Q, Q_bar, sigma, S, hard=False)
    x_ref = od3pm.p_sample(logits, x, t, tn, Q, Q_bar, sigma, S, e=e)
    dq, qb = _step_mats(Q, Q_bar, sigma, t, tn)
    p = rt.reverse_step_d3pm(logits.cuda(), x.cuda(), dq, qb, S, hard=False).cpu()
    torch.testing.assert_close(p, p_ref, rtol=1e-5, atol=1e-12)
    xh = rt.reverse_step_d3pm(logits.cuda(), x.cuda(), dq, qb, S, hard=True, noise=e.view(N, Lq, S)).cpu()
    v = p_ref / e.view(N, Lq, S)
    top2 = v.topk(2, dim=-1).values
    ok = (top2[..., 0] - top2[..., 1]) > 1e-4 * top2[..., 0]
    assert ok.float().mean() > 0.9 and torch.equal(xh[ok], x_ref[ok])


@pytest.mark.parametrize('name', ['d3pm_small', 'd3pm_real'])
def test_q_sample_d3pm_bit_exact(rt, name):
    g = load_golden(name)
    Q, Q_bar, sigma = _sched()
    x = g['x']
    out = rt.q_sample_d3pm(x.cuda(), g['tq'], Q_bar.cuda().contiguous(), S, noise=g['eq'].view(x.shape[0], x.shape[1], S))
    assert torch.equal(out.cpu(), g['q_sample'])


@pytest.mark.parametrize('cfg,B,L', [(BN_DEV_SMALL, 3, 20), (dict(), 2, 389), (BN_DEV_SMALL, 3, 389),
                                     (dict(n_layers=9), 3, 389)])   # M >= 1024: persistent GEMM + conv taps
def test_bytenet_forward_vs_oracle(rt, cfg, B, L):
    from oracle import bytenet as obn
    from sgpo_amd import weights as W
    sd = W.bytenet_synthetic_state(seed=5, **cfg)
    net = rt.BytenetDenoiser(sd)
    g = torch.Generator().manual_seed(2)
    x = torch.randint(0, S, (B, L), generator=g)
    for t in (7, 480):
        ref, href = obn.bytenet_forward(sd, x, torch.tensor([t] * B), return_hidden=True)
        got, hid = net.forward(x.cuda(), t, return_hidden=True)
        eh = (hid.cpu() - href).abs().max() / href.abs().max()
        el = (got.cpu() - ref).abs().max() / ref.abs().max()
        print(f'bytenet d_model={net.d_model} L={L} t={t}: hidden err {eh:.3e} logits err {el:.3e}')
        assert eh < BN_TOL and el < BN_TOL
    # per-sequence timesteps
    tt = torch.tensor([3, 100, 499][:B])
    ref = obn.bytenet_forward(sd, x, tt)
    got = net.forward(x.cuda(), tt).cpu()
    assert (got - ref).abs().max() / ref.abs().max() < BN_TOL


def test_bytenet_forward_large_batch_matches_small_batch(rt):
    """M = 13 x 389 = 5057 rows: every GEMM (1x1, dilated conv, fp32 output and fp32 residual) takes the 256x256
    tile of the persistent kernel with a ragged last tile; a sequence's logits must not depend on the batch it
    is in (the 3-sequence batch runs the 256x128 tile)."""
    from sgpo_amd import weights as W
    sd = W.bytenet_synthetic_state(seed=5, n_layers=9)
    net = rt.BytenetDenoiser(sd)
    g = torch.Generator().manual_seed(4)
    x = torch.randint(0, S, (13, 389), generator=g).cuda()
    tt = torch.randint(1, 500, (13,), generator=g).cuda()
    big, hbig = net.forward(x, tt, return_hidden=True)
    for sl in (slice(0, 3), slice(10, 13)):
        small, hsmall = net.forward(x[sl].contiguous(), tt[sl].contiguous(), return_hidden=True)
        assert torch.equal(hsmall, hbig[sl]) and torch.equal(small, big[sl])


def test_bytenet_rows_do_not_depend_on_the_batch_size(rt):
    """Below the size where the fused LayerNorm GEMM takes over (42 sequences of 389), every GEMM path (64- and 128-row
    kernel, persistent 256x128 / 256x256 by round count, 1x1 and dilated-conv forms) gives a sequence the same logits
    whatever batch it is in."""
    from sgpo_amd import weights as W
    sd = W.bytenet_synthetic_state(seed=5, n_layers=9)
    net = rt.BytenetDenoiser(sd)
    g = torch.Generator().manual_seed(8)
    x = torch.randint(0, S, (40, 389), generator=g).cuda()
    tt = torch.randint(1, 500, (40,), generator=g).cuda()
    first = net.forward(x[:1].contiguous(), tt[:1].contiguous())
    for B in (2, 3, 5, 6, 11, 21, 22, 40):
        out = net.forward(x[:B].contiguous(), tt[:B].contiguous())
        last = net.forward(x[B - 1:B].contiguous(), tt[B - 1:B].contiguous())
        assert torch.equal(out[:1], first) and torch.equal(out[B - 1:B], last), B


def _launch_counts(fn):
    """{kernel kind: launches} of the library's own launch profiler (sgpo_prof_*) around fn()."""
    import ctypes as C
    from sgpo_amd import _lib as L
    lib = L.load()
    n = 64
    L.check(lib.sgpo_prof_enable(1, 4096), 'prof_enable')
    lib.sgpo_prof_reset()
    out = fn()
    torch.cuda.synchronize()
    ms, cnt, work = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)()
    L.check(lib.sgpo_prof_read(n, ms, cnt, work), 'prof_read')
    lib.sgpo_prof_enable(0, 0)
    return out, {lib.sgpo_prof_kind_name(k).decode(): int(cnt[k]) for k in range(n) if cnt[k]}


def test_bytenet_fused_layernorm_gemm_vs_two_pass_and_oracle(rt, monkeypatch):
    """M = 44 x 389 = 17116 rows (ragged last 128-row tile): the PFF1 and dilated-conv GEMMs take the full-row persistent
    kernel whose epilogue does bias + LayerNorm + GELU (gemm_p.hip EPI_LN_GELU).  Checked against the CPU restatement,
    and against the two-pass form (SGPO_BYTENET_FUSED_LN=0: fp32-output GEMM + ln_gelu_row512_kernel), which since round 3
    runs the epilogue's arithmetic operation for operation: the two must agree BIT FOR BIT (a shard
    below the fused kernel's row threshold has to reproduce the rows of the unsharded batch)."""
    from oracle import bytenet as obn
    from sgpo_amd import weights as W
    sd = W.bytenet_synthetic_state(seed=5, n_layers=3)
    net = rt.BytenetDenoiser(sd)
    g = torch.Generator().manual_seed(6)
    B, L = 44, 389
    x = torch.randint(0, S, (B, L), generator=g)
    tt = torch.randint(1, 500, (B,), generator=g)
    (got, hid), k_fused = _launch_counts(lambda: net.forward(x.cuda(), tt.cuda(), return_hidden=True))
    monkeypatch.setenv('SGPO_BYTENET_FUSED_LN', '0')
    (got2, hid2), k_pass = _launch_counts(lambda: net.forward(x.cuda(), tt.cuda(), return_hidden=True))
    monkeypatch.delenv('SGPO_BYTENET_FUSED_LN')
    again, _ = net.forward(x.cuda(), tt.cuda(), return_hidden=True)
    assert torch.equal(again, got)                                     # deterministic
    fused_kind = 'gemm_bf16_kernel<EPI_LN_GELU>'
    assert k_fused.get(fused_kind, 0) == 2 * 3 and fused_kind not in k_pass, (k_fused, k_pass)   # the fused kernel really ran
    assert torch.equal(hid, hid2) and torch.equal(got, got2), \
        f'fused vs two-pass: hidden {(hid - hid2).abs().max() / hid2.abs().max():.3e} of the range'
    ref, href = obn.bytenet_forward(sd, x[:6], tt[:6], return_hidden=True)
    eh = (hid[:6].cpu() - href).abs().max() / href.abs().max()
    el = (got[:6].cpu() - ref).abs().max() / ref.abs().max()
    print(f'vs oracle: hidden {eh:.3e} logits {el:.3e} of the range')
    assert eh < BN_TOL and el < BN_TOL


def test_bytenet_rows_do_not_depend_on_the_fused_kernel_threshold(rt):
    """`gemm_ln_gelu` takes the fused EPI_LN_GELU kernel from (M + 127) / 128 >= n_cu / 2 tiles on
    (about 43 sequences of 389 on 256 CUs) and the GEMM + pass below.  One sequence's logits across that switch
    (B = 40 vs 44 vs 64) and a 64-sequence call against its two 32-sequence shards: bit-identical."""
    from sgpo_amd import weights as W
    sd = W.bytenet_synthetic_state(seed=5, n_layers=3)
    net = rt.BytenetDenoiser(sd)
    g = torch.Generator().manual_seed(9)
    x = torch.randint(0, S, (64, 389), generator=g).cuda()
    tt = torch.randint(1, 500, (64,), generator=g).cuda()
    fused_kind = 'gemm_bf16_kernel<EPI_LN_GELU>'
    whole, k64 = _launch_counts(lambda: net.forward(x, tt))
    b40, k40 = _launch_counts(lambda: net.forward(x[:40].contiguous(), tt[:40].contiguous()))
    b44, k44 = _launch_counts(lambda: net.forward(x[:44].contiguous(), tt[:44].contiguous()))
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    if n_cu == 256:
        assert fused_kind in k64 and fused_kind in k44 and fused_kind not in k40, (k40, k44, k64)
    assert torch.equal(b40, whole[:40]) and torch.equal(b44, whole[:44])
    lo = net.forward(x[:32].contiguous(), tt[:32].contiguous())
    hi = net.forward(x[32:].contiguous(), tt[32:].contiguous())
    assert torch.equal(torch.cat([lo, hi]), whole)


def _mh_setup(rt, g):
    from sgpo_amd import weights as W
    res1 = [int(r) for r in g['residues']]
    msd = W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=len(res1))
    vf = rt.MlpValueFunction(msd, res1)              # reads the raw 1-indexed list (defect D3)
    return vf, [r - 1 for r in res1], msd, res1


@pytest.mark.parametrize('name', ['daps_small', 'daps_real'])
def test_metropolis_hastings_kernel_against_reference(rt, name):
    g = load_golden(name)
    vf, residues0, msd, res1 = _mh_setup(rt, g)
    L = g['x0hat'].shape[1]
    for tag in ('lo', 'hi', 'inf'):
        s = torch.as_tensor(g[f'sigma_{tag}']) / 1000
        a = (1 - torch.exp(-s)) * (1 - 1 / S)
        la, l1a = float(torch.log(a + 1e-5)), float(torch.log(1 - a))
        out = rt.daps_mh(vf, g['x0hat'].cuda(), residues0, S, la, l1a, 128., g['r_idx'].shape[0], r_idx=g['r_idx'],
                         r_val=g['r_val'], r_u=g['r_u'])
        assert torch.equal(out.cpu(), g[f'mh_{tag}']), tag
    # zero steps returns x0hat
    out = rt.daps_mh(vf, g['x0hat'].cuda(), residues0, S, -1.0, -0.1, 128., 0)
    assert torch.equal(out.cpu(), g['x0hat'])


@pytest.mark.parametrize('N,Lq,res1,steps', [(1, 5, [3], 40), (70, 30, [1, 2, 30], 200), (3, 389, [117, 118, 119, 162, 166, 301], 500)])
def test_mh_shape_sweep(rt, N, Lq, res1, steps):
    """One chain on one residue, more chains than a workgroup holds, the last position of the sequence as a residue:
    chains with injected proposals against the restatement, token for token."""
    from oracle import daps as odaps
    from oracle import mlp as omlp
    from sgpo_amd import weights as W
    msd = W.mlp_synthetic_state(seed=60 + N, n_residues=len(res1))
    vf = rt.MlpValueFunction(msd, res1)                     # reads the raw 1-indexed list (defect D3)
    pred = omlp.Predictor(msd, res1)
    residues0 = [r - 1 for r in res1]
    gen = torch.Generator().manual_seed(N + steps)
    x0 = torch.randint(0, S, (N, Lq + 1), generator=gen)    # one extra column: the raw list may name position Lq
    r_idx = torch.randint(len(residues0), (steps, N), generator=gen)
    r_val = torch.randint(S, (steps, N), generator=gen)
    r_u = torch.rand(steps, N, generator=gen)
    for sigma, alpha in ((torch.tensor(0.5), 128.), (torch.tensor(float('inf')), 4.)):
        ref = odaps.metropolis_hasting(x0, pred.log_likelihood, sigma, steps, alpha, 1000, residues0, S, Lq + 1, r_idx, r_val, r_u)
        a = (1 - torch.exp(-sigma / 1000)) * (1 - 1 / S)
        la, l1a = float(torch.log(a + 1e-5)), float(torch.log(1 - a))
        out = rt.daps_mh(vf, x0.cuda(), residues0, S, la, l1a, alpha, steps, r_idx=r_idx, r_val=r_val, r_u=r_u)
        assert torch.equal(out.cpu(), ref)
    assert (ref != x0).any()


def test_mh_longer_chain_vs_oracle_and_philox(rt):
    """1000-step chains: injected noise vs the oracle; Philox mode only moves the mutable residues."""
    from oracle import daps as odaps
    from oracle import mlp as omlp
    g = load_golden('daps_small')
    vf, residues0, msd, res1 = _mh_setup(rt, g)
    N, L, steps = 16, g['x0hat'].shape[1], 1000
    gen = torch.Generator().manual_seed(3)
    x0 = torch.randint(0, S, (N, L), generator=gen)
    r_idx = torch.randint(len(residues0), (steps, N), generator=gen)
    r_val = torch.randint(S, (steps, N), generator=gen)
    r_u = torch.rand(steps, N, generator=gen)
    sigma = torch.tensor(float('inf'))
    pred = omlp.Predictor(msd, res1)
    ref = odaps.metropolis_hasting(x0, pred.log_likelihood, sigma, steps, 4.0, 1000, residues0, S, L, r_idx, r_val, r_u)
    a = (1 - torch.exp(-sigma / 1000)) * (1 - 1 / S)
    la, l1a = float(torch.log(a + 1e-5)), float(torch.log(1 - a))
    out = rt.daps_mh(vf, x0.cuda(), residues0, S, la, l1a, 4.0, steps, r_idx=r_idx, r_val=r_val, r_u=r_u).cpu()
    # accept decisions compare u with exp(fp32 difference): allow a few near-tie chains to diverge
    same = (out == ref).all(dim=1).float().mean().item()
    assert same >= 0.8, same
    assert (ref != x0).any()
    ph = rt.daps_mh(vf, x0.cuda(), residues0, S, la, l1a, 4.0, steps, noise_src=rt.NoiseSource(seed=11)).cpu()
    frozen = [l for l in range(L) if l not in residues0]
    assert torch.equal(ph[:, frozen], x0[:, frozen]) and (ph != x0).any()
    ph2 = rt.daps_mh(vf, x0.cuda(), residues0, S, la, l1a, 4.0, steps, noise_src=rt.NoiseSource(seed=11)).cpu()
    assert torch.equal(ph, ph2)


def test_d3pm_dropin_and_daps_loop(rt):
    from models.d3pm import D3PM
    from oracle import bytenet as obn
    from oracle import d3pm as od3pm
    from problem.protein import ProteinPredictor
    from sampling.daps import DAPS
    from sampling.uncond import Uncond
    from sgpo_amd import weights as W
    g = load_golden('daps_small')
    Lq = g['x0hat'].shape[1]
    res1 = [int(r) for r in g['residues']]
    full = bytes(g['full_seq'].numpy().tolist()).decode()
    sd = W.bytenet_synthetic_state(seed=6, **BN_DEV_SMALL)
    net = D3PM('synthetic', Lq, device='cuda:0', state_dict=sd)
    Q, Q_bar, sigma = _sched()
    assert torch.equal(net.sigma.cpu(), sigma) and net.S == 26 and net.timestep == 500
    # teacher-forced p_sample against the oracle posterior on the oracle's fp32 logits
    gen = torch.Generator().manual_seed(4)
    x = torch.randint(0, S, (3, Lq), generator=gen)
    from tf_common import TeacherForced
    tf, i_step = TeacherForced(logit_rtol=2e-2, atol=1e-6), 0
    for t, tn in ((499, 472), (250, 237), (26, 0)):
        e = torch.empty(3, Lq, S, dtype=torch.float64).exponential_(1, generator=gen)
        logits = obn.bytenet_forward(sd, x, torch.tensor([t] * 3))
        p_ref = od3pm.p_sample(logits, x, t, tn, Q, Q_bar, sigma, S, hard=False)
        x_ref = od3pm.p_sample(logits, x, t, tn, Q, Q_bar, sigma, S, e=e.view(-1, S))
        p = net.p_sample(x.cuda(), torch.tensor(t), torch.tensor(tn), hard=False).cpu()
        net.injected_noise.append(e)
        xo = net.p_sample(x.cuda(), torch.tensor(t), torch.tensor(tn)).cpu()
        # the bf16 bound is on LOGITS (1e-2 of the largest reference logit); the posterior takes p0 = softmax(z) SQUARED
        # (defect D9, models/d3pm.py:255-257), so a logit error counts twice: logit_rtol 2e-2 in tf_common's bookkeeping,
        # tokens compared wherever the race margin exceeds twice that, fraction printed and asserted by finish()
        tf.step(i_step, p, p_ref, xo, x_ref.view(3, Lq), e, float(logits[..., :S].abs().max()))
        i_step += 1
    tf.finish('d3pm drop-in p_sample')
    # full DAPS loop through the drop-in API
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq=full, residues=res1)
    pred = ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0')
    pred.model.load_state_dict(W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=len(res1)))
    pred.eval()
    algo = DAPS(net, forward_op=pred, num_steps=4, ode_steps=5, mh_steps=200, alpha=8., beta=1000, data_config=dc,
                device='cuda:0')
    net.reseed(21)
    x1, seqs = algo.inference(num_samples=32, verbose=False, detokenize=True)
    assert x1.shape == (32, Lq) and x1.dtype == torch.int64 and len(seqs) == 32
    frozen = [l for l in range(Lq) if l not in algo.residues]
    full_tok = torch.tensor([ALPHABET.index(c) for c in full])
    assert torch.equal(x1.cpu()[:, frozen], full_tok[frozen].expand(32, -1))      # projection (daps.py:154)
    assert int(x1.max()) < S and int(x1.min()) >= 0
    net.reseed(21)
    x2 = algo.inference(num_samples=32, verbose=False)
    assert torch.equal(x1, x2)
    # unguided D3PM ancestral sampling (uncond.py:56-62) on a short horizon
    un = Uncond(net, None, device='cuda:0', data_config=dc)
    net.timestep = 12
    xu = un.inference(num_samples=8, verbose=False)
    net.timestep = 500
    assert xu.shape == (8, Lq) and int(xu.max()) < S


def test_cls_guidance_on_d3pm_generic_step(rt):
    """Classifier guidance with a D3PM net (configs/model/d3pm.yaml + cls_guidance_discrete.yaml) has no fused kernel:
    the step is p_sample(soft, fp64) -> -1 at x -> D3PM.get_guided_rates (fused jump ratios) -> +1 -> clamp ->
    normalise -> sample.  Teacher-forced against the CPU restatement with the same Exp(1) draws; tokens compared
    where the sampling margin exceeds the bf16 denoiser tolerance."""
    from models.d3pm import D3PM
    from oracle import bytenet as obn
    from oracle import cls_guidance as ocg
    from oracle import d3pm as od3pm
    from oracle import mlp as omlp
    from problem.protein import ProteinPredictor
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    from sgpo_amd import weights as W
    Lq, B, temp = 24, 4, 0.5
    res1 = [2, 5, 6, 11, 17, 20]
    sd = W.bytenet_synthetic_state(seed=8, **BN_DEV_SMALL)
    net = D3PM('synthetic', Lq, device='cuda:0', state_dict=sd)
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq='M' * Lq, residues=res1)
    pred = ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0')
    msd = W.mlp_synthetic_state(seed=3, n_residues=len(res1))
    pred.model.load_state_dict(msd)
    pred.eval()
    algo = Classifier_Guidance_Inpaint(net, dc, forward_op=pred, temperature=temp, device='cuda:0')
    Q, Q_bar, sigma = _sched()
    ref_pred = omlp.Predictor(msd, res1)
    gen = torch.Generator().manual_seed(12)
    x = torch.randint(0, S, (B, Lq), generator=gen)
    agree = []
    for t in (480, 250, 30):
        e = torch.empty(B, Lq, S).exponential_(1, generator=gen)
        logits = obn.bytenet_forward(sd, x, torch.tensor([t] * B))
        q = od3pm.p_sample(logits, x, t, None, Q, Q_bar, sigma, S, hard=False)

        def ratio_fn(qm):
            r = torch.zeros(B, Lq, S)
            base = ref_pred(x, torch.full((B,), float(t))).squeeze(-1)
            for d in range(Lq):
                for s in range(S):
                    xj = x.clone()
                    xj[:, d] = s
                    r[:, d, s] = ref_pred(xj, torch.full((B,), float(t))).squeeze(-1) - base
            return qm * torch.exp(torch.clamp(r / temp, max=80))
        x_ref, p_ref = ocg.cg_step(q, x, ratio_fn, e.double())
        net.injected_noise.append(e)
        x_dev = algo._generic_step(x.cuda(), torch.tensor(t)).cpu()
        v = p_ref / e
        top2 = v.topk(2, dim=-1).values
        ok = (top2[..., 0] - top2[..., 1]) > 1e-1 * top2[..., 0]
        agree.append(float(ok.float().mean()))
        assert torch.equal(x_dev[ok], x_ref[ok]), t
    assert min(agree) > 0.7
    # the whole loop through the drop-in API: 499 steps, frozen positions projected back
    net.reseed(2)
    xs = algo.inference(num_samples=8)
    frozen = [l for l in range(Lq) if l not in algo.residues]
    assert xs.shape == (8, Lq) and bool((xs[:, frozen] == ALPHABET.index('M')).all()) and int(xs.max()) < S


@pytest.mark.parametrize('tag', ['unguided', 'guided'])
def test_cls_guidance_on_d3pm_teacher_forced_against_reference_trajectory(rt, tag):
    """Classifier_Guidance_Inpaint on a D3PM net -- the round-0 data path of pareto.py:48-67 (temperature 0) and the
    guided form -- walked over the REFERENCE's own 499-step run (golden cg_d3pm_small, written by tools/refgen
    gen_cg_d3pm; the restatement reproduces it state by state, tests/test_oracle_d3pm.py): at every 6th state and the
    last 12 the drop-in's step gives the oracle's probabilities within the bf16 logit bound (squared p0: logit_rtol
    2e-2, as in test_d3pm_dropin_and_daps_loop) and the reference's next tokens wherever the race margin exceeds it."""
    from conftest import golden_noise_d3pm
    from models.d3pm import D3PM
    from oracle import bytenet as obn
    from oracle import cls_guidance as ocg
    from oracle import d3pm as od3pm
    from oracle import mdlm as omdlm
    from oracle import mlp as omlp
    from problem.protein import ProteinPredictor
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    from sgpo_amd import weights as W
    from tf_common import TeacherForced
    g = load_golden('cg_d3pm_small')
    B, Lq = g[f'x_{tag}'].shape
    T, temp = int(g['T']), float(g[f'temp_{tag}'])
    res1 = [int(r) for r in g['residues']]
    full = bytes(g['full_seq'].numpy().tolist()).decode()
    sd = W.bytenet_synthetic_state(seed=int(g['bn_seed']), **BN_DEV_SMALL)
    net = D3PM('synthetic', Lq, device='cuda:0', state_dict=sd)
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq=full, residues=res1)
    msd = W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=len(res1))
    pred = ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0')
    pred.model.load_state_dict(msd)
    pred.eval()
    algo = Classifier_Guidance_Inpaint(net, dc, forward_op=pred if temp else None, temperature=temp, device='cuda:0')
    opred = omlp.Predictor(msd, res1) if temp else None
    Q, Q_bar, sigma = _sched()
    x0, es = golden_noise_d3pm(g, B, Lq)
    states = g[f'states_{tag}'].long()
    assert torch.equal(states[0], x0)
    timesteps = torch.linspace(T - 1, 1, T - 1, dtype=int)
    tf = TeacherForced(logit_rtol=2e-2, atol=1e-6)
    picks = sorted(set(range(0, T - 1, 6)) | set(range(T - 13, T - 1)))
    for i in picks:
        t, xs, e = timesteps[i], states[i], es[i]
        logits = obn.bytenet_forward(sd, xs, torch.tensor([int(t)] * B))
        q = od3pm.p_sample(logits, xs, int(t), None, Q, Q_bar, sigma, S, hard=False)
        x_ref, p_ref = ocg.cg_step(q, xs, lambda qm: omdlm.get_guided_rates(opred, xs, t, qm, S, guide_temp=temp)[0], e)
        if i + 1 < T - 1:
            assert torch.equal(x_ref, states[i + 1]), f'oracle left the reference trajectory at step {i}'
        net.injected_noise.append(e)
        xo, p = algo._generic_step(xs.cuda(), t, want_probs=True)
        tf.step(i, p.cpu().double(), p_ref, xo.cpu(), x_ref, e, float(logits[..., :S].abs().max()))
    tf.finish(f'cg_d3pm_small/{tag}')
