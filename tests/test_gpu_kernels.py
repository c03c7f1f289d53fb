"""GPU parity tests of the individual HIP kernels (through the C ABI) against the CPU oracle
and the goldens captured from the reference.

Tolerances: fp32 kernels (value function, reverse step) rtol 1e-4 as `north_star` states;
the bf16-MFMA denoiser rtol 1e-2 against the fp32 oracle (defect D7: the reference's own GPU
path is bf16 autocast) and tighter against the oracle's bf16-emulating mode.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

DIM_KEYS = ('hidden', 'cond_dim', 'n_blocks', 'n_heads')


@pytest.fixture(scope='module')
def rt():
    from sgpo_amd import runtime
    runtime.init('cuda:0')
    return runtime


def _sd(g):
    from sgpo_amd import weights as W
    dims = dict(zip(DIM_KEYS, [int(v) for v in g['dims']]))
    return W.dit_synthetic_state(seed=int(g['seed']), vocab=31, **dims)


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (256, 1536, 512), (119 * 3, 512, 2048), (1000, 31, 512),
                                   (130, 192, 128),
                                   # M >= 1024: the persistent 256x128 kernel (gemm_p.hip), incl. ragged M/N tails
                                   (3000, 512, 512), (2125, 1536, 512), (4096, 2048, 64), (2500, 512, 2048),
                                   (1500, 31, 512), (70000, 256, 128)])
def test_gemm_bf16(rt, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    out = rt.gemm_bf16(a, w)
    ref = a.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().T
    torch.testing.assert_close(out.double(), ref, rtol=1e-4, atol=1e-4)


def test_gemm_identity_asymmetric(rt):
    # A = I against an asymmetric W catches swapped row/col maps in the MFMA epilogue
    K = 128
    a = torch.eye(K).cuda()
    w = (torch.arange(192 * K, dtype=torch.float32).reshape(192, K) % 251 - 125).cuda()
    out = rt.gemm_bf16(a, w)
    assert torch.equal(out, w.T.contiguous())


@pytest.mark.parametrize('name', ['dit_small_s0', 'dit_small_s1', 'dit_real_s0', 'dit_real_s1'])
def test_dit_forward(rt, name):
    from oracle import dit as odit
    g = load_golden(name)
    sd = _sd(g)
    net = rt.DitDenoiser(sd)
    x = g['x'].cuda()
    sigma = g['sigma'].cuda()
    logits, hidden = net.forward(x, sigma, return_hidden=True)
    logits, hidden = logits.cpu(), hidden.cpu()
    # (a) against the reference's fp32 CPU logits (golden): bf16 tolerance
    scale = g['logits'].abs().max()
    err = (logits - g['logits']).abs().max() / scale
    assert err < 1e-2, f'max err / max |logit| = {err:.3e}'      # north_star: bf16 rtol 1e-2
    errh = (hidden - g['h_last']).abs().max() / g['h_last'].abs().max()
    print(f'{name}: last hidden state, max err / max |h| = {errh:.3e}')
    assert errh < 1e-2, f'hidden state: {errh:.3e}'                # the same bf16 bound on the residual stream
    # (b) the device error must be of the size the bf16 autocast rounding itself produces:
    #     compare with the oracle rounding to bf16 at the points torch autocast does (defect D7)
    ref16 = odit.dit_forward(sd, g['x'], g['sigma'], emulate_bf16=True)
    err_emu = (ref16 - g['logits']).abs().max() / scale
    err16 = (logits - ref16).abs().max() / scale
    print(f'{name}: dev-vs-fp32 {err:.3e}  bf16emu-vs-fp32 {err_emu:.3e}  dev-vs-bf16emu {err16:.3e}')
    assert err < 2.0 * err_emu + 2e-3
    assert err16 < 1e-2
    # x_emb bypass (dit.py:397-410)
    l2 = net.forward(x, sigma, x_emb=(g['h_last'] * 0.5 + 0.1).cuda()).cpu()
    err2 = (l2 - g['logits_xemb']).abs().max() / g['logits_xemb'].abs().max()
    assert err2 < 1e-2, f'x_emb path: {err2:.3e}'


@pytest.mark.parametrize('dims,B,L', [(dict(hidden=128, cond_dim=32, n_blocks=2, n_heads=2), 3, 389),
                                      (dict(hidden=512, cond_dim=128, n_blocks=2, n_heads=8), 2, 389),
                                      (dict(hidden=128, cond_dim=32, n_blocks=1, n_heads=2), 2, 161),
                                      (dict(hidden=128, cond_dim=32, n_blocks=1, n_heads=2), 3, 20),
                                      (dict(hidden=128, cond_dim=32, n_blocks=1, n_heads=2), 2, 33),
                                      (dict(hidden=128, cond_dim=32, n_blocks=1, n_heads=2), 2, 320),
                                      (dict(hidden=128, cond_dim=32, n_blocks=1, n_heads=2), 2, 321),
                                      (dict(hidden=128, cond_dim=32, n_blocks=1, n_heads=2), 1, 512),
                                      (dict(hidden=128, cond_dim=32, n_blocks=1, n_heads=2), 2, 700),
                                      (dict(hidden=128, cond_dim=32, n_blocks=1, n_heads=2), 1, 1024),
                                      (dict(hidden=256, cond_dim=64, n_blocks=2, n_heads=4), 2, 50)])   # a width between the two with goldens
def test_dit_forward_long_sequences(rt, dims, B, L):
    """TrpB length (389) and the lengths around the attention kernel's variants: a single key tile (20), idle
    waves (33), the first length past the 4-wave kernel (161: K/V images > 40 KB switch to 8 waves per workgroup),
    the last length with K/V resident in LDS and the first that streams them through the ring (320 / 321), two full
    rounds of query tiles (512), lengths past what the resident kernels could hold at all (700: 22 tiles, a ragged
    third round; 1024: the default max_len).  Device logits against the CPU restatement rounding to bf16 where
    autocast does."""
    from oracle import dit as odit
    from sgpo_amd import weights as W
    sd = W.dit_synthetic_state(seed=9, vocab=31, **dims)
    net = rt.DitDenoiser(sd)
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 31, (B, L), generator=g)
    sigma = torch.rand(B, generator=g) * 3
    logits = net.forward(x.cuda(), sigma.cuda()).cpu()
    ref16 = odit.dit_forward(sd, x, sigma, emulate_bf16=True)
    ref32 = odit.dit_forward(sd, x, sigma)
    scale = ref32.abs().max()
    err, err_emu = (logits - ref32).abs().max() / scale, (ref16 - ref32).abs().max() / scale
    print(f'L={L} hidden={dims["hidden"]}: dev-vs-fp32 {err:.3e}  bf16emu-vs-fp32 {err_emu:.3e}')
    assert err < 2.0 * err_emu + 2e-3 and (logits - ref16).abs().max() / scale < 1.5e-2
    # a sequence's logits do not depend on its neighbours in the batch
    one = net.forward(x[B - 1:B].cuda(), sigma[B - 1:B].cuda()).cpu()
    assert torch.equal(one[0], logits[B - 1])


@pytest.mark.parametrize('per_seq', [True, False])
def test_dit_forward_fused_residual_layernorm_is_bit_identical(rt, monkeypatch, per_seq):
    """M = 1075 x 119 = 127925 rows (1000 tiles of 128 rows, the last one ragged): the out-projection and fc2 GEMMs take the
    full-row persistent kernel whose epilogue also does the LayerNorm-modulate that follows (gemm_p.hip EPI_GATE_RES_LN;
    the launch counters confirm it ran).  The separate pass (SGPO_DIT_FUSED_LN=0) and the fused kernel share one summation
    order (csrc/ln_stats.h): every logit and every hidden value must be equal bit for bit -- with one modulation row per
    sequence and with one for the whole batch -- and rows of the big batch must equal the same sequences run as a small
    batch (64-row GEMM + separate pass)."""
    import ctypes as C
    from sgpo_amd import _lib as L
    from sgpo_amd import weights as W
    lib = L.load()
    sd = W.dit_synthetic_state(seed=21)
    net = rt.DitDenoiser(sd)
    g = torch.Generator().manual_seed(10)
    B, Lq = 1075, 119
    x = torch.randint(0, 31, (B, Lq), generator=g).cuda()
    sigma = (torch.rand(B, generator=g) * 3).cuda() if per_seq else torch.full((1,), 0.7).cuda()

    def fused_launches(fn):
        L.check(lib.sgpo_prof_enable(1, 256), 'prof_enable')
        out = fn()
        torch.cuda.synchronize()
        n = L.PROF_KINDS
        ms, cnt, work = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)()
        L.check(lib.sgpo_prof_read(n, ms, cnt, work), 'prof_read')
        lib.sgpo_prof_enable(0, 0)
        by_name = {lib.sgpo_prof_kind_name(k).decode(): int(cnt[k]) for k in range(n)}
        return out, by_name.get('gemm_bf16_kernel<EPI_GATE_RES_LN>', 0)

    (fused, hf), n_fused = fused_launches(lambda: net.forward(x, sigma, return_hidden=True))
    monkeypatch.setenv('SGPO_DIT_FUSED_LN', '0')
    (plain, hp), n_plain = fused_launches(lambda: net.forward(x, sigma, return_hidden=True))
    monkeypatch.delenv('SGPO_DIT_FUSED_LN')
    assert n_fused == 16 and n_plain == 0
    assert torch.isfinite(fused).all()
    assert torch.equal(hf, hp) and torch.equal(fused, plain)
    small = net.forward(x[1070:1073].contiguous(), sigma[1070:1073].contiguous() if per_seq else sigma)
    assert torch.equal(small, fused[1070:1073])


def test_dit_forward_rows_do_not_depend_on_the_batch_size(rt):
    """Every GEMM path the launcher can pick -- the 64- and 128-row kernel, the persistent kernel with 256x128 and with
    256x256 tiles (chosen by round count), the full-row kernel with the fused LayerNorm (from 512 tiles at >= 75 % fill) --
    must give a sequence the same logits, bit for bit, whatever batch it sits in.  Batch sizes on both sides of each
    switch (M = B x 119): 8 | 9 (1024 rows), 17 | 18 and 34 | 35 (a quarter of the CUs' worth of tiles for N = 1536 / 512),
    100, 400 (256x128 wins on rounds), 540 | 560 (511 | 521 full-row tiles), 820."""
    from sgpo_amd import weights as W
    sd = W.dit_synthetic_state(seed=22)
    net = rt.DitDenoiser(sd)
    g = torch.Generator().manual_seed(12)
    Lq, Bmax = 119, 820
    x = torch.randint(0, 31, (Bmax, Lq), generator=g).cuda()
    sigma = (torch.rand(Bmax, generator=g) * 3).cuda()
    first = net.forward(x[:1].contiguous(), sigma[:1].contiguous())
    for B in (8, 9, 17, 18, 34, 35, 64, 100, 137, 200, 400, 540, 560, 820):
        out = net.forward(x[:B].contiguous(), sigma[:B].contiguous())
        last = net.forward(x[B - 1:B].contiguous(), sigma[B - 1:B].contiguous())
        assert torch.equal(out[:1], first), B
        assert torch.equal(out[B - 1:B], last), B


@pytest.mark.parametrize('B', [12, 36])     # M = 1428: 256x128 tiles only; M = 4284: 256x256 tiles with a ragged last tile
def test_dit_forward_large_batch_uses_persistent_gemm(rt, B):
    """M = B*L >= 1024 routes every GEMM (all epilogues: RoPE, GELU, gate+residual, vocab head) through
    the persistent kernel; rows are independent, so the result must match the oracle like the small case."""
    from oracle import dit as odit
    from sgpo_amd import weights as W
    sd = W.dit_synthetic_state(seed=21)
    net = rt.DitDenoiser(sd)
    g = torch.Generator().manual_seed(9)
    Lq = 119
    x = torch.randint(0, 31, (B, Lq), generator=g)
    sigma = torch.rand(B, generator=g) * 3
    logits, hidden = net.forward(x.cuda(), sigma.cuda(), return_hidden=True)
    ref, hs = odit.dit_forward(sd, x, sigma, return_hidden_states=True)
    scale = ref.abs().max()
    err = (logits.cpu() - ref).abs().max() / scale
    eh = (hidden.cpu() - hs[-1]).abs().max() / hs[-1].abs().max()
    print(f'persistent-GEMM DiT: logits err {err:.3e} hidden err {eh:.3e}')
    assert err < 2e-2 and eh < 2e-2
    # the first two sequences alone (M = 238 -> 128x128 kernel) agree with their rows in the big batch
    #    bit for bit (shared epilogue arithmetic): sharding a batch never changes its samples
    small = net.forward(x[:2].cuda(), sigma[:2].cuda())
    assert torch.equal(small, logits[:2])
    tail = net.forward(x[-2:].cuda(), sigma[-2:].cuda())          # rows of the ragged last tile
    assert torch.equal(tail, logits[-2:])


def test_dit_sigma_broadcast(rt):
    g = load_golden('dit_small_s0')
    net = rt.DitDenoiser(_sd(g))
    x = g['x'].cuda()
    a = net.forward(x, None)
    b = net.forward(x, torch.zeros(1).cuda())
    c = net.forward(x, torch.zeros(x.shape[0]).cuda())
    assert torch.equal(a, b) and torch.equal(a, c)


def _margin_ok(p, den, tok, tol=1e-4):
    """rows where the race winner beats the runner-up by more than tol (relative)."""
    v = p / den
    top2 = v.topk(2, dim=-1).values
    return (top2[..., 0] - top2[..., 1]) > tol * top2[..., 0]


def test_reverse_step_mdlm_soft_and_hard(rt):
    from oracle import mdlm as omdlm
    from sgpo_amd import _lib as L
    g = load_golden('mdlm_small')
    T = int(g['T'])
    x = g['x']
    B = x.shape[0]
    # the oracle's own fp32 logits feed both sides: isolates the fused kernel
    from oracle import dit as odit
    logits = odit.dit_forward(_sd(g), x, torch.zeros(B))
    for t in (127, 64, 1):
        mt, ms, _ = omdlm.move_chances(t, T, B)
        mtf, msf = float(mt[0, 0, 0]), float(ms[0, 0, 0])
        _, q = rt.reverse_step_mdlm(logits.cuda(), x.cuda(), mtf, msf, 28, L.STEP_SOFT)
        torch.testing.assert_close(q.cpu(), g[f'soft_t{t}'], rtol=1e-4, atol=1e-7)
        u = g[f'u_t{t}']
        xh, _ = rt.reverse_step_mdlm(logits.cuda(), x.cuda(), mtf, msf, 28, L.STEP_HARD, noise=u.cuda())
        ok = _margin_ok(g[f'soft_t{t}'], 1e-10 - (u + 1e-10).log(), None) | (x != 28)
        assert ok.float().mean() > 0.95
        assert torch.equal(xh.cpu()[ok], g[f'hard_t{t}'][ok])


@pytest.mark.parametrize('B,Lq,K', [(1, 1, 2), (3, 7, 5), (2, 64, 26), (17, 119, 31), (2, 389, 32), (1, 600, 31), (5, 33, 31)])
def test_reverse_step_shape_sweep(rt, B, Lq, K):
    """Ragged and extreme shapes (one position, two letters, the full 32-letter limit, lengths off every tile size):
    MDLM soft / hard and UDLM soft / hard against the restatement on the same random logits and injected noise."""
    from oracle import mdlm as omdlm
    from sgpo_amd import _lib as L
    g = torch.Generator().manual_seed(B * 1000 + Lq * 10 + K)
    mask = K - 1 if K < 31 else 28
    logits = torch.randn(B, Lq, K, generator=g) * 3
    x = torch.randint(0, K, (B, Lq), generator=g)
    x[torch.rand(B, Lq, generator=g) < 0.5] = mask
    u = torch.rand(B, Lq, K, generator=g)
    T, t = 128, 77
    mt, ms, _ = omdlm.move_chances(t, T, B)
    mtf, msf = float(mt[0, 0, 0]), float(ms[0, 0, 0])
    logp = omdlm.subs_parameterization(logits, x, mask)
    q_ref = omdlm.p_sample(None, x, t, T, mask, hard=False, log_p_x0=logp)
    x_ref = omdlm.p_sample(None, x, t, T, mask, hard=True, u=u, log_p_x0=logp)
    _, q = rt.reverse_step_mdlm(logits.cuda(), x.cuda(), mtf, msf, mask, L.STEP_SOFT)
    torch.testing.assert_close(q.cpu(), q_ref, rtol=1e-4, atol=1e-7)
    xh, _ = rt.reverse_step_mdlm(logits.cuda(), x.cuda(), mtf, msf, mask, L.STEP_HARD, noise=u.cuda())
    ok = _margin_ok(q_ref, 1e-10 - (u + 1e-10).log(), None) | (x != mask)
    assert torch.equal(xh.cpu()[ok], x_ref[ok]) and ok.float().mean() > 0.9
    assert torch.equal(xh.cpu()[x != mask], x[x != mask])                       # carry-over
    # UDLM (uniform diffusion, d3pm parameterisation)
    logp_u = omdlm.d3pm_parameterization(logits, mask)
    qu_ref = omdlm.p_sample(None, x, t, T, mask, diffusion='uniform', hard=False, log_p_x0=logp_u)
    xu_ref = omdlm.p_sample(None, x, t, T, mask, diffusion='uniform', hard=True, u=u, log_p_x0=logp_u)
    _, qu = rt.reverse_step_udlm(logits.cuda(), x.cuda(), mtf, msf, L.STEP_SOFT)
    torch.testing.assert_close(qu.cpu(), qu_ref, rtol=2e-4, atol=1e-7)
    xuh, _ = rt.reverse_step_udlm(logits.cuda(), x.cuda(), mtf, msf, L.STEP_HARD, noise=u.cuda())
    oku = _margin_ok(qu_ref, 1e-10 - (u + 1e-10).log(), None)
    assert torch.equal(xuh.cpu()[oku], xu_ref[oku]) and oku.float().mean() > 0.9


def test_reverse_step_udlm(rt):
    from oracle import dit as odit
    from oracle import mdlm as omdlm
    from sgpo_amd import _lib as L
    g = load_golden('udlm_small')
    T = int(g['T'])
    x = g['x']
    B = x.shape[0]
    for t in (127, 3):
        mt, ms, sig = omdlm.move_chances(t, T, B)
        logits = odit.dit_forward(_sd(g), x, sig)
        mtf, msf = float(mt[0, 0, 0]), float(ms[0, 0, 0])
        _, q = rt.reverse_step_udlm(logits.cuda(), x.cuda(), mtf, msf, L.STEP_SOFT)
        torch.testing.assert_close(q.cpu(), g[f'soft_t{t}'], rtol=2e-4, atol=1e-7)
        u = g[f'u_t{t}']
        xh, _ = rt.reverse_step_udlm(logits.cuda(), x.cuda(), mtf, msf, L.STEP_HARD, noise=u.cuda())
        ok = _margin_ok(g[f'soft_t{t}'], 1e-10 - (u + 1e-10).log(), None)
        assert ok.float().mean() > 0.95
        assert torch.equal(xh.cpu()[ok], g[f'hard_t{t}'][ok])


def test_q_sample_mdlm(rt):
    from oracle import mdlm as omdlm
    g = load_golden('mdlm_small')
    T = int(g['T'])
    mc = 1 - torch.exp(-omdlm.total_noise(g['tq'] / T))
    out = rt.q_sample_mdlm(g['x_clean'].cuda(), mc.cuda(), 28, noise=g['uq'].cuda())
    assert torch.equal(out.cpu(), g['q_sample'])


def test_mlp_forward_and_jump_ratios(rt):
    from oracle import mdlm as omdlm
    from oracle import mlp as omlp
    from sgpo_amd import weights as W
    for name, n_res in (('mlp_creilov', 119), ('mlp_trpb', 15)):
        g = load_golden(name)
        sd = W.mlp_synthetic_state(seed=int(g['seed']), n_residues=n_res)
        residues = [int(r) for r in g['residues']] if 'residues' in g else None
        read_pos = residues if residues is not None else list(range(g['x'].shape[1]))
        vf = rt.MlpValueFunction(sd, read_pos)
        y = vf.forward(g['x'].cuda(), g['t'].cuda()).cpu()
        torch.testing.assert_close(y, g['y'].squeeze(-1), rtol=1e-4, atol=1e-5)
        ll = vf.forward(g['x'].cuda(), None).cpu()
        torch.testing.assert_close(ll, g['ll'], rtol=1e-4, atol=1e-5)
        # jump log-ratios against the materialised reference formulation (oracle)
        S = 31 if residues is None else 26
        xs = g['x'][:2]
        pred = omlp.Predictor(sd, residues)
        t = 17.0
        lp = pred(xs, t * torch.ones(2))
        xj = omdlm.get_all_jump_transitions(xs, S)
        lpj = pred(xj, t * torch.ones(xj.shape[0])).view(2, xs.shape[1], S)
        ref = (lpj - lp.view(2, 1, 1))[:, read_pos, :]
        got = vf.jump_log_ratios(xs.cuda(), t, S).cpu()
        torch.testing.assert_close(got, ref, rtol=1e-4, atol=2e-5)
        # positions the value function does not read: ratio 0 up to BLAS batch-order rounding in the reference
        if residues is not None:
            others = [d for d in range(xs.shape[1]) if d not in read_pos]
            assert float((lpj - lp.view(2, 1, 1))[:, others, :].abs().max()) < 1e-5


@pytest.mark.parametrize('B,Lq,read_pos,S', [(1, 1, [0], 2), (2, 9, [8], 31), (3, 40, [0, 39], 26), (1, 389, list(range(0, 389, 26)), 26),
                                             (70, 23, list(range(23)), 31), (5, 64, list(range(64)), 20)])
def test_value_function_shape_sweep(rt, B, Lq, read_pos, S):
    """One residue, one sequence, a batch that is no multiple of the tile, read positions at both ends of the
    sequence: f(x, t), f(x) and the S x n_read jump log-ratios against the materialised formulation of the reference."""
    from oracle import mdlm as omdlm
    from oracle import mlp as omlp
    from sgpo_amd import weights as W
    sd = W.mlp_synthetic_state(seed=7 + Lq, n_residues=len(read_pos))
    vf = rt.MlpValueFunction(sd, read_pos)
    pred = omlp.Predictor(sd, read_pos)
    g = torch.Generator().manual_seed(B + Lq)
    x = torch.randint(0, S, (B, Lq), generator=g)
    t = torch.randint(0, 128, (B,), generator=g).float()
    torch.testing.assert_close(vf.forward(x.cuda(), t.cuda()).cpu(), pred(x, t).squeeze(-1), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(vf.forward(x.cuda(), None).cpu(), pred.log_likelihood(x), rtol=1e-4, atol=1e-5)
    tt = 17.0
    lp = pred(x, tt * torch.ones(B))
    xj = omdlm.get_all_jump_transitions(x, S)
    lpj = pred(xj, tt * torch.ones(xj.shape[0])).view(B, Lq, S)
    ref = (lpj - lp.view(B, 1, 1))[:, read_pos, :]
    got = vf.jump_log_ratios(x.cuda(), tt, S).cpu()
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=2e-5)
    assert float(got[torch.arange(B)[:, None], torch.arange(len(read_pos))[None, :], x[:, read_pos]].abs().max()) == 0.0   # s = x_d


def test_absorbing_jump_ratios_skip_only_zero_rate_entries(rt):
    """sgpo_guidance_ratio_mlp_absorbing: bit-identical to the full kernel at masked positions and in the
    mask column; 0 elsewhere; and the guided step (whose prior is 0 at the skipped entries) gives the
    same probabilities and tokens bit for bit."""
    from sgpo_amd import _lib as L
    from sgpo_amd import weights as W
    MASK, K = 28, 31
    g = torch.Generator().manual_seed(5)
    for Lq, read_pos, B in ((119, list(range(119)), 37), (40, [1, 5, 6, 17, 33, 39], 9), (70, list(range(70)), 64)):
        msd = W.mlp_synthetic_state(seed=3, n_residues=len(read_pos))
        vf = rt.MlpValueFunction(msd, read_pos)
        x = torch.randint(0, 26, (B, Lq), generator=g)
        frac = torch.rand(B, 1, generator=g)                 # from fully masked to fully unmasked rows
        frac[0], frac[1] = 1.0, 0.0
        x[torch.rand(B, Lq, generator=g) < frac] = MASK
        xd = x.cuda()
        full = vf.jump_log_ratios(xd, 5.0, K).clone()
        comp = vf.jump_log_ratios(xd, 5.0, K, mask_index=MASK).clone()
        masked = (x[:, read_pos] == MASK).cuda()
        assert torch.equal(comp[masked], full[masked])
        assert torch.equal(comp[..., MASK], full[..., MASK])
        other = torch.ones(K, dtype=torch.bool); other[MASK] = False
        assert float(comp[~masked][:, other.cuda()].abs().max()) == 0.0
        if len(read_pos) == Lq:
            logits = torch.randn(B, Lq, K, generator=g).cuda()
            e = torch.empty(B, Lq, K).exponential_(1, generator=g).cuda()
            a = rt.reverse_step_mdlm(logits, xd, 0.6, 0.3, MASK, L.STEP_SOFT_CG, noise=e, log_ratio=full, guide_temp=0.064,
                                     want_probs=True)
            b = rt.reverse_step_mdlm(logits, xd, 0.6, 0.3, MASK, L.STEP_SOFT_CG, noise=e, log_ratio=comp, guide_temp=0.064,
                                     want_probs=True)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_cg_step_against_golden(rt):
    """One guided step: q -> (q - 1_x) * ratio + 1_x -> clamp -> normalise (cls_guidance.py:38-51)."""
    from oracle import cls_guidance as ocg
    from oracle import dit as odit
    from oracle import mdlm as omdlm
    from oracle import mlp as omlp
    from sgpo_amd import _lib as L
    from sgpo_amd import weights as W
    g = load_golden('cg_small_guided')
    temp = float(g['temp'])
    xs = g['xs']
    B, Lq = xs.shape
    T = int(g['T'])
    msd = W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=Lq)
    vf = rt.MlpValueFunction(msd, list(range(Lq)))
    lr = vf.jump_log_ratios(xs.cuda(), 9.0, 31)
    logits = odit.dit_forward(_sd(g), xs, torch.zeros(B))
    mt, ms, _ = omdlm.move_chances(9, T, B)
    e = g['es'][0]
    xo, p = rt.reverse_step_mdlm(logits.cuda(), xs.cuda(), float(mt[0, 0, 0]), float(ms[0, 0, 0]), 28, L.STEP_SOFT_CG,
                                 noise=e.cuda(), log_ratio=lr, guide_temp=temp, want_probs=True)
    pred = omlp.Predictor(msd)
    x_ref, p_ref = ocg.cg_step(g['q'], xs, lambda qm: omdlm.get_guided_rates(pred, xs, 9, qm, 31, temp)[0], e)
    torch.testing.assert_close(p.cpu(), p_ref, rtol=2e-4, atol=1e-6)
    ok = _margin_ok(p_ref, e, None)
    assert ok.float().mean() > 0.9
    assert torch.equal(xo.cpu()[ok], x_ref[ok])


def test_guided_step_flags_nan_rates_and_rows_it_cannot_sample(rt):
    """models/mdlm.py:410-411 raises on NaN guided rates and torch.multinomial (cls_guidance.py:52) on a row whose sum is
    0, inf or NaN: the kernel reports both through nan_flag (bits 0 and 1) and the drop-in raises at the step."""
    from sgpo_amd import _lib as L
    MASK, K, B, Lq = 28, 31, 3, 10
    g = torch.Generator().manual_seed(2)
    logits = torch.randn(B, Lq, K, generator=g).cuda()
    x = torch.randint(0, 26, (B, Lq), generator=g)
    x[:, ::2] = MASK
    x = x.cuda()
    e = torch.empty(B, Lq, K).exponential_(1, generator=g)
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    lr = torch.zeros(B, Lq, K, device='cuda')
    rt.reverse_step_mdlm(logits, x, 0.6, 0.3, MASK, L.STEP_SOFT_CG, noise=e, log_ratio=lr, guide_temp=0.5, nan_flag=flag)
    assert int(flag.item()) == 0
    lr[1, 4, 7] = float('nan')                                    # a NaN log-ratio at a masked position
    rt.reverse_step_mdlm(logits, x, 0.6, 0.3, MASK, L.STEP_SOFT_CG, noise=e, log_ratio=lr, guide_temp=0.5, nan_flag=flag)
    assert int(flag.item()) & 1
    flag.zero_()
    bad = logits.clone()
    bad[2, 6] = float('inf')                                      # unguided: the row's probabilities are NaN
    rt.reverse_step_mdlm(bad, x, 0.6, 0.3, MASK, L.STEP_SOFT_CG, noise=e, nan_flag=flag)
    assert int(flag.item()) == 2
    # through the drop-in: raised at the step, and the flag is cleared for the next call
    import sgpo_amd
    sgpo_amd.install_dropin()
    from models.mdlm import MDLM
    from problem.protein import ProteinPredictor
    from sgpo_amd import weights as W
    dims = dict(hidden=128, cond_dim=32, n_blocks=1, n_heads=2)
    net = MDLM('synthetic', dict(model=dict(hidden_size=128, cond_dim=32, n_blocks=1, n_heads=2)), None, num_steps=16,
               seq_len=Lq, device='cuda:0', state_dict=W.dit_synthetic_state(seed=1, vocab=31, **dims))
    dc = dict(alphabet='ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!', alphabet_size=31, seq_len=Lq, full_seq='M' * Lq, residues=None)
    pred = ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0')
    msd = W.mlp_synthetic_state(seed=4, n_residues=Lq)
    msd['cls_head.2.weight'] = torch.full((1, 64), float('nan'))  # f = NaN for every jump -> every evaluated ratio NaN
    pred.model.load_state_dict(msd)
    pred.eval()
    with pytest.raises(ValueError, match='NaNs'):
        net.guided_step(x, torch.tensor(9), pred, 0.5)
    pred.model.load_state_dict(W.mlp_synthetic_state(seed=4, n_residues=Lq))
    pred.eval()
    net.guided_step(x, torch.tensor(9), pred, 0.5)                 # healthy again


def test_philox_sampler_statistics_and_shard_invariance(rt):
    from sgpo_amd.runtime import NoiseSource, categorical_sample
    K = 31
    p = torch.rand(K, generator=torch.Generator().manual_seed(1))
    p = p / p.sum()
    rows = 200000
    probs = p.repeat(rows, 1).cuda()
    full = categorical_sample(probs, NoiseSource(seed=7))
    freq = torch.bincount(full.cpu(), minlength=K).float() / rows
    assert (freq - p).abs().max() < 5e-3
    # two shards with global row offsets reproduce the unsharded draw bit for bit
    h = rows // 2
    a = categorical_sample(probs[:h], NoiseSource(seed=7, row_offset=0))
    b = categorical_sample(probs[h:], NoiseSource(seed=7, row_offset=h))
    assert torch.equal(torch.cat([a, b]), full)
    # a different call offset gives a different stream
    ns = NoiseSource(seed=7)
    ns.next_offset()
    c = categorical_sample(probs, ns)
    assert (c != full).float().mean() > 0.5
