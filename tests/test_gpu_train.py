"""GPU parity of row f-3: the lockstep ensemble trainer through the C ABI against the
weights the reference's own train_classifier produced (tests/golden/train_mlp.npz) and against the CPU
restatement on several members with different batches."""
import pytest
import torch

from conftest import load_golden
from oracle import train_mlp
from test_oracle_train import ORDER, case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def rt():
    from sgpo_amd import runtime
    runtime.init('cuda:0')
    return runtime


def _close(got, want):
    # fp32 sums in another order; Adam divides by sqrt(v) + 1e-8, so a gradient entry of the order of 1e-8 can move
    # its weight by a visible fraction of lr = 1e-3: bound the worst entry loosely and the bulk tightly
    for k in ORDER:
        d = (got[k].cpu() - want[k]).abs()
        assert float(d.max()) < 2e-4, (k, float(d.max()))
        assert float(d.mean()) < 2e-6, (k, float(d.mean()))


def test_trainer_against_reference_weights(rt):
    g = load_golden('train_mlp')
    for name in ('full', 'sites', 'clean'):
        sd0, batches, residues, sd1, losses = case(g, name)
        L = batches[0][0].shape[1]
        read = residues if residues is not None else list(range(L))
        tr = rt.MlpEnsembleTrainer([sd0, sd0], read)          # two identical members: identical results
        got_losses = []
        for xt, t, y in batches:
            x2 = torch.stack([xt, xt]).cuda()
            l = tr.step(x2, torch.stack([t, t]).float(), torch.stack([y, y]))
            got_losses.append(l.cpu())
        got_losses = torch.stack(got_losses)
        assert torch.equal(got_losses[:, 0], got_losses[:, 1])
        torch.testing.assert_close(got_losses[:, 0], losses, rtol=2e-4, atol=1e-6)
        sds = tr.state_dicts()
        for k in ORDER:
            assert torch.equal(sds[0][k], sds[1][k])
        _close(sds[0], sd1)
        assert torch.equal(sds[0]['time_embedder.0.W'].cpu(), sd0['time_embedder.0.W'])


def test_trainer_members_with_their_own_batches(rt):
    from sgpo_amd import weights as W
    gen = torch.Generator().manual_seed(8)
    Lq, n_res, E, steps = 119, 119, 5, 12
    states = [W.mlp_synthetic_state(seed=40 + e, n_residues=n_res) for e in range(E)]
    tr = rt.MlpEnsembleTrainer(states, list(range(Lq)))
    per_member = [[] for _ in range(E)]
    for s in range(steps):
        B = 64 if s % 4 != 3 else 17                            # the ragged last batch of an epoch
        x = torch.randint(0, 31, (E, B, Lq), generator=gen)
        t = torch.randint(0, 128, (E, B), generator=gen)
        y = torch.randn(E, B, generator=gen)
        loss = tr.step(x.cuda(), t.float(), y).cpu()
        for e in range(E):
            per_member[e].append((x[e], t[e], y[e], float(loss[e])))
    sds = tr.state_dicts()
    for e in range(E):
        want, losses = train_mlp.train(states[e], [(a, b, c) for a, b, c, _ in per_member[e]])
        torch.testing.assert_close(torch.tensor([q[3] for q in per_member[e]]), torch.tensor(losses), rtol=5e-4, atol=1e-6)
        _close(sds[e], want)
    # the trained weights drive the sampling-side value function
    vf = rt.MlpValueFunction(sds[0], list(range(Lq)))
    x = torch.randint(0, 31, (9, Lq), generator=gen)
    from oracle import mlp as omlp
    ref = omlp.mlp_forward({k: v.cpu() for k, v in sds[0].items()}, x, torch.full((9,), 5.0)).squeeze(-1)
    torch.testing.assert_close(vf.forward(x.cuda(), torch.full((9,), 5.0)).cpu(), ref, rtol=1e-4, atol=1e-5)


def test_trainer_several_steps_in_one_call(rt):
    """`sgpo_mlp_trainer_steps`: n batches in one library call give the weights and losses of n single-step calls, bit for bit."""
    from sgpo_amd import weights as W
    gen = torch.Generator().manual_seed(9)
    Lq, E, n, B = 119, 3, 7, 64
    states = [W.mlp_synthetic_state(seed=60 + e, n_residues=Lq) for e in range(E)]
    x = torch.randint(0, 31, (n, E, B, Lq), generator=gen).cuda()
    t = torch.randint(0, 128, (n, E, B), generator=gen).float().cuda()
    y = torch.randn(n, E, B, generator=gen).cuda()
    one, many = rt.MlpEnsembleTrainer(states, list(range(Lq))), rt.MlpEnsembleTrainer(states, list(range(Lq)))
    losses = torch.stack([one.step(x[j], t[j], y[j]) for j in range(n)])
    assert torch.equal(many.steps(x, t, y), losses)
    for a, b in zip(one.state_dicts(), many.state_dicts()):
        assert all(torch.equal(a[k], b[k]) for k in a)


def test_trainer_rejects_bad_shapes(rt):
    from sgpo_amd import weights as W
    sd = W.mlp_synthetic_state(seed=1, n_residues=10)
    tr = rt.MlpEnsembleTrainer([sd], list(range(10)))
    with pytest.raises(Exception):
        tr.step(torch.zeros(1, 65, 10, dtype=torch.int64).cuda(), torch.zeros(1, 65), torch.zeros(1, 65))   # B > 64
    with pytest.raises(Exception):
        rt.MlpEnsembleTrainer([W.mlp_synthetic_state(seed=1, n_residues=11)], list(range(10)))


def test_train_classifier_dropin(rt, tmp_path):
    """training.train_classifier drop-in: same call as the reference, members trained in lockstep, learnable target."""
    import sgpo_amd
    sgpo_amd.install_dropin()
    from models.mdlm import MDLM
    from problem.protein import ProteinPredictor
    from training.train_classifier import train_classifier, train_classifier_ensemble
    from sgpo_amd import weights as W
    ALPHABET = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'
    Lq = 24
    cfg = dict(diffusion='absorbing_state', parameterization='subs', time_conditioning=False,
               model=dict(hidden_size=128, cond_dim=32, n_blocks=1, n_heads=2, length=64))
    net = MDLM('synthetic', cfg, None, num_steps=128, seq_len=Lq, device='cuda:0',
               state_dict=W.dit_synthetic_state(seed=0, hidden=128, cond_dim=32, n_blocks=1, n_heads=2))
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq='A' * Lq, residues=None)
    gen = torch.Generator().manual_seed(0)
    n = 150
    X = torch.randint(0, 20, (n, Lq), generator=gen)
    w = torch.randn(Lq, 20, generator=gen) * 0.3
    Y = w[torch.arange(Lq)[None, :], X].sum(dim=1)                       # additive landscape
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=64, shuffle=True)
    tc = dict(learning_rate=1e-3, wandb=False, n_epochs=60, n_epochs_no_time=60, batch_size=64)

    def fresh(k):
        torch.manual_seed(100 + k)
        return ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0')

    def mse(c):
        c.eval()
        with torch.no_grad():
            return float(((c.log_likelihood(X.cuda()) - Y.cuda()) ** 2).mean())
    clfs = [fresh(k) for k in range(3)]
    before = [mse(c) for c in clfs]
    g1 = torch.Generator(device='cuda:0').manual_seed(5)
    train_classifier_ensemble(clfs, net, loader, tc, save_dir=str(tmp_path), time_conditioned=False, generator=g1)
    after = [mse(c) for c in clfs]
    assert all(a < 0.25 * b for a, b in zip(after, before)), (before, after)
    assert (tmp_path / 'classifier_2.pt').exists()
    # reproducible under the same generator seed, and different members stay different
    clfs2 = [fresh(k) for k in range(3)]
    g2 = torch.Generator(device='cuda:0').manual_seed(5)
    train_classifier_ensemble(clfs2, net, loader, tc, time_conditioned=False, generator=g2)
    for a, b in zip(clfs, clfs2):
        for (ka, va), (kb, vb) in zip(a.model.state_dict().items(), b.model.state_dict().items()):
            assert torch.equal(va, vb), ka
    assert not torch.equal(clfs[0].model.mlp[0].weight, clfs[1].model.mlp[0].weight)
    # time-conditioned path (q_sample noising through the MDLM drop-in) and the single-member signature
    c = fresh(9)
    b0 = mse(c)
    tc2 = dict(tc, n_epochs=30)
    out = train_classifier(c, net, loader, tc2, save_dir=str(tmp_path / 'single'), project_fn=None, ensemble_idx=4)
    assert out is c and (tmp_path / 'single' / 'classifier_4.pt').exists()
    assert mse(c) < 0.6 * b0
