"""GPU parity of the NOS step (MDLM._nos_denoise) against the golden captured from the reference
and against the autograd oracle (delta after the Adagrad steps)."""
import pytest
import torch

from conftest import load_golden

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


@pytest.mark.parametrize('name', ['nos_small', 'nos_real'])   # nos_real: BASELINE config 4 at full size (real DiT dims, L = 389)
def test_nos_denoise_against_reference(env, name):
    from models.mdlm import MDLM
    from oracle import nos as onos
    from problem.protein_nos import NOSHead
    from sgpo_amd import weights as W
    g = load_golden(name)
    dims = dict(zip(DIM_KEYS, [int(v) for v in g['dims']]))
    sd = W.dit_synthetic_state(seed=int(g['seed']), vocab=31, **dims)
    cfg = dict(model=dict(hidden_size=dims['hidden'], cond_dim=dims['cond_dim'], n_blocks=dims['n_blocks'],
                          n_heads=dims['n_heads']))
    B, Lq = g['x'].shape
    T = int(g['T'])
    net = MDLM('synthetic', cfg, None, num_steps=T, seq_len=Lq, device='cuda:0', state_dict=sd)
    head = NOSHead(dims['hidden'])
    with torch.no_grad():
        head.output_layer.weight.copy_(g['w_cls'])
        head.output_layer.bias.copy_(g['b_cls'])
    head = head.cuda().eval()
    for tag in bytes(g['tags'].numpy().tolist()).decode():
        t, nsteps, lr, coef = [float(v) for v in g[f'hp_{tag}']]
        u = g[f'u_{tag}']
        net.injected_noise.append(u)
        xs, probs, _ = net._nos_denoise(head, int(nsteps), lr, coef, g['x'].cuda(), torch.tensor(int(t)))
        p_ref = g[f'probs_{tag}']
        # Tolerance.  north_star's bf16 figure (logits within 1e-2) is NOT reachable for this step, for a reason that
        # lies in the algorithm: Adagrad's first update is lr * g / sqrt(g^2) = +-lr whatever the size of g, so a
        # coordinate of delta whose gradient is near zero moves by a full +-nos_step_size in a direction decided by
        # rounding; a bf16 backbone (the reference's own GPU arithmetic, defect D7) flips some of those signs against
        # the fp32 CPU run the golden holds.  Two checks instead: (i) an absolute bound at the measured size
        # (real dims: relative error of p up to 6e-2 at stability 1, 1.6e-1 at stability 200, |dp| <= 1.2e-2);
        # (ii) the device is no further from the fp32 reference than ~2x what the CPU restatement itself moves when
        # only its backbone pass rounds to bf16 where autocast does -- i.e. the error is the bf16 backbone's, not
        # the kernels'.  The fp32 machinery itself (closed-form backward + Adagrad from the same fp32 hidden state)
        # is held to 2e-3 / 1e-3 further down.
        from oracle import dit as odit
        from sgpo_amd import runtime as rt
        _, _, d_ref = onos.nos_denoise(sd, g['w_cls'], g['b_cls'], g['x'], int(t), T, 28, int(nsteps), lr, coef, u)
        _, hs = odit.dit_forward(sd, g['x'], torch.zeros(B), return_hidden_states=True)
        _, p_emu, _ = onos.nos_denoise(sd, g['w_cls'], g['b_cls'], g['x'], int(t), T, 28, int(nsteps), lr, coef, u,
                                       emulate_bf16=True)
        rtol_p, atol_p = 5e-2, 2e-3
        dp, de = (probs.cpu() - p_ref).abs(), (p_emu - p_ref).abs()
        big = p_ref > 1e-3
        print(f'[nos {name} {tag}] device vs fp32 reference: max |dp| {float(dp.max()):.2e}, worst relative error where '
              f'p > 1e-3 {float((dp[big] / p_ref[big]).max()):.2e}; bf16-backbone restatement vs fp32 reference: max |dp| '
              f'{float(de.max()):.2e}, worst relative {float((de[big] / p_ref[big]).max()):.2e}')
        torch.testing.assert_close(probs.cpu(), p_ref, rtol=rtol_p, atol=atol_p)
        assert float(dp.max()) < 2.0 * float(de.max()) + 1e-4
        # tokens must agree wherever the race p / E cannot flip inside that tolerance: the reference winner's lower
        # bound beats every other entry's upper bound (unmasked positions carry over whatever the probabilities)
        E = 1e-10 - (u + 1e-10).log()
        lo, hi = (p_ref * (1 - rtol_p) - atol_p).clamp_min(0) / E, (p_ref * (1 + rtol_p) + atol_p) / E
        win = (p_ref / E).argmax(-1, keepdim=True)
        ok = (lo.gather(-1, win) > hi.scatter(-1, win, 0.).max(-1, keepdim=True).values).squeeze(-1) | (g['x'] != 28)
        print(f'{name} {tag}: {float(ok.float().mean()):.3f} of the positions are decided beyond the tolerance')
        assert ok.float().mean() > (0.8 if name == 'nos_small' else 0.6)
        assert torch.equal(xs.cpu()[ok], g[f'xs_{tag}'][ok]), tag
        assert torch.equal(xs.cpu()[g['x'] != 28], g['x'][g['x'] != 28])      # carry-over
        # the optimised delta itself against the autograd oracle, fed the oracle's fp32 hidden state so
        # that only the closed-form backward + Adagrad are compared (fp32 both sides)
        mt, ms, _ = net._move_chances(int(t))
        xs2, p2, d = rt.nos_denoise(net.net, g['x'].cuda(), g['w_cls'], int(nsteps), lr, coef, mt, ms, 28,
                                    want_delta=True, noise=u, hidden_in=hs[-1])
        scale = float(d_ref.abs().max())
        err = float((d.cpu() - d_ref).abs().max()) / scale
        print(f'nos {tag}: max|delta| {scale:.3e}, rel err vs autograd oracle {err:.3e}')
        assert err < 2e-3
        torch.testing.assert_close(p2.cpu(), p_ref, rtol=1e-3, atol=1e-6)
        assert torch.equal(xs2.cpu()[ok], g[f'xs_{tag}'][ok])


def test_nos_hidden_256_against_oracle(env):
    """The third supported width (128 and 512 have goldens): closed-form backward + Adagrad on the device against the
    autograd restatement, both from the restatement's fp32 hidden state."""
    from models.mdlm import MDLM
    from oracle import dit as odit
    from oracle import nos as onos
    from sgpo_amd import runtime as rt
    from sgpo_amd import weights as W
    dims = dict(hidden=256, cond_dim=64, n_blocks=1, n_heads=4)
    sd = W.dit_synthetic_state(seed=23, vocab=31, **dims)
    cfg = dict(model=dict(hidden_size=256, cond_dim=64, n_blocks=1, n_heads=4))
    B, Lq, T = 3, 45, 128
    net = MDLM('synthetic', cfg, None, num_steps=T, seq_len=Lq, device='cuda:0', state_dict=sd)
    g = torch.Generator().manual_seed(8)
    x = torch.randint(0, 26, (B, Lq), generator=g)
    x[:, ::2] = 28
    x[2, :] = 7                                              # a row without masked positions
    w_cls, b_cls = torch.randn(1, 256, generator=g) * 0.5, torch.randn(1, generator=g)
    u = torch.rand(B, Lq, 31, generator=g)
    t, nsteps, lr, coef = 90, 4, 0.5, 50.0
    xs_ref, p_ref, d_ref = onos.nos_denoise(sd, w_cls, b_cls, x, t, T, 28, nsteps, lr, coef, u)
    _, hs = odit.dit_forward(sd, x, torch.zeros(B), return_hidden_states=True)
    mt, ms, _ = net._move_chances(t)
    xs, p, d = rt.nos_denoise(net.net, x.cuda(), w_cls, nsteps, lr, coef, mt, ms, 28, want_delta=True, noise=u, hidden_in=hs[-1])
    assert float((d.cpu() - d_ref).abs().max()) / float(d_ref.abs().max()) < 2e-3
    torch.testing.assert_close(p.cpu(), p_ref, rtol=1e-3, atol=1e-6)
    E = 1e-10 - (u + 1e-10).log()
    top2 = (p_ref / E).topk(2, dim=-1).values
    ok = ((top2[..., 0] - top2[..., 1]) > 1e-2 * top2[..., 0]) | (x != 28)
    assert ok.float().mean() > 0.9 and torch.equal(xs.cpu()[ok], xs_ref[ok])
    assert torch.equal(xs.cpu()[2], x[2])
    # no inner steps: delta stays 0 and the call is the plain guided reverse step from the unshifted hidden state
    _, p0_ref, _ = onos.nos_denoise(sd, w_cls, b_cls, x, t, T, 28, 0, lr, coef, u)
    _, p0, d0 = rt.nos_denoise(net.net, x.cuda(), w_cls, 0, lr, coef, mt, ms, 28, want_delta=True, noise=u, hidden_in=hs[-1])
    assert float(d0.abs().max()) == 0.0
    torch.testing.assert_close(p0.cpu(), p0_ref, rtol=1e-4, atol=1e-6)
    # every inner step of an unmasked row has the same gradient -w_cls / L: its delta is a function of the column
    # only (the kernel forms it once per wave), so the rows of sequence 2 and the unmasked rows of the others agree
    assert torch.equal(d[2, 0], d[2, -1]) and torch.equal(d[0, 1], d[2, 0])


def test_nos_sampler_loop(env):
    from models.mdlm import MDLM
    from problem.protein_nos import NOSHead
    from sampling.NOS import NOS
    from sgpo_amd import weights as W
    dims = dict(hidden=128, cond_dim=32, n_blocks=2, n_heads=2)
    sd = W.dit_synthetic_state(seed=2, vocab=31, **dims)
    cfg = dict(model=dict(hidden_size=128, cond_dim=32, n_blocks=2, n_heads=2))
    Lq, T = 30, 12
    net = MDLM('synthetic', cfg, None, num_steps=T, seq_len=Lq, device='cuda:0', state_dict=sd)
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq='M' * Lq, residues=[3, 4, 5, 10, 11, 20])
    head = NOSHead(128).cuda().eval()
    algo = NOS(net, dc, forward_op=head, num_nos_steps=3, nos_step_size=0.5, nos_stability_coef=0.1, device='cuda:0')
    net.reseed(5)
    x, seqs = algo.inference(num_samples=16, verbose=False, detokenize=True)
    assert x.shape == (16, Lq) and len(seqs) == 16
    frozen = [l for l in range(Lq) if l not in algo.residues]
    assert bool((x[:, frozen] == ALPHABET.index('M')).all())
    net.reseed(5)
    assert torch.equal(x, algo.inference(num_samples=16, verbose=False))
    # unguided branch (nos_stability_coef None -> p_sample hard, NOS.py:34-36)
    algo.nos_stability_coef = None
    xu = algo.inference(num_samples=4, verbose=False)
    assert xu.shape == (4, Lq)
