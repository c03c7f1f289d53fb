"""GPU parity for the BASELINE configurations the round-1 suite left out: the residue-subset (TrpB) form of the fused
guided step, config 1 (prior_sample.py: temperature 0, CreiLOV, B = 64) and config 5 (iterativeBO.py: a trained
ensemble member guides batches of 10).  Reference trajectories come from tests/golden/ (made by importing the
reference, tools/refgen/make_goldens.py); the CPU oracle supplies the per-step probabilities at those states."""
import numpy as np
import pytest
import torch

from conftest import golden_noise, load_golden
from tf_common import ALPHABET, DIM_KEYS, TeacherForced, oracle_soft_step

pytestmark = pytest.mark.gpu

CREILOV = ('MAGLRHTFVVADATLPDCPLVYASEGFYAMTGYGPDEVLGHNARFLQGEGTDPKEVQKIRDAIKKGEACSVRLLNYRKDGTPFWNLL'
           'TVTPIKTPDGRVSKFVGVQVDVTSKTEGKALA')


@pytest.fixture(scope='module')
def env():
    import sgpo_amd
    sgpo_amd.install_dropin()
    from sgpo_amd import runtime
    runtime.init('cuda:0')
    return runtime


def _mdlm(g, T, Lq):
    from models.mdlm import MDLM
    from sgpo_amd import weights as W
    dims = dict(zip(DIM_KEYS, [int(v) for v in g['dims']]))
    sd = W.dit_synthetic_state(seed=int(g['seed']), vocab=31, **dims)
    cfg = dict(diffusion='absorbing_state', parameterization='subs', time_conditioning=False,
               model=dict(hidden_size=dims['hidden'], cond_dim=dims['cond_dim'], n_blocks=dims['n_blocks'],
                          n_heads=dims['n_heads'], length=1024))
    return MDLM('synthetic', cfg, None, num_steps=T, seq_len=Lq, device='cuda:0', state_dict=sd), sd


def test_cg_teacher_forced_trpb_residue_subset(env):
    """Every TrpB run of MDLM + cls_guidance: 15 designable residues, the value function reads the raw 1-indexed list
    (defect D3), so the fused step goes through the slot_of_pos map and the compact jump tiles.  Walks the REFERENCE's
    own states (golden `states`), real DiT dims, L = 389, temperature 0.064."""
    from oracle import cls_guidance as ocg
    from oracle import mdlm as omdlm
    from oracle import mlp as omlp
    from problem.protein import ProteinPredictor
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    from sgpo_amd import weights as W
    g = load_golden('cg_real_guided_trpb')
    T, temp = int(g['T']), float(g['temp'])
    B, Lq = g['x'].shape
    res = [int(r) for r in g['residues']]
    full_seq = bytes(g['full_seq'].numpy().tolist()).decode()
    es = golden_noise(g, B, Lq)
    net, sd = _mdlm(g, T, Lq)
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq=full_seq, residues=res)
    pred = ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0')
    msd = W.mlp_synthetic_state(seed=int(g['mlp_seed']), n_residues=len(res))
    pred.model.load_state_dict(msd)
    pred.eval()
    algo = Classifier_Guidance_Inpaint(net, dc, forward_op=pred, temperature=temp, device='cuda:0')
    opred = omlp.Predictor(msd, res)
    states = g['states'].long()
    timesteps = torch.linspace(T - 1, 1, T - 1, dtype=int)
    tf = TeacherForced()
    x_last = None
    for i, t in enumerate(timesteps):
        xs = states[i]
        q, zmax = oracle_soft_step(sd, xs, t, T)
        x_ref, p_ref = ocg.cg_step(q, xs, lambda qm: omdlm.get_guided_rates(opred, xs, t, qm, 31, temp)[0], es[i])
        if i + 1 < T - 1:
            assert torch.equal(x_ref, states[i + 1])            # the oracle walks the reference's trajectory
        net.injected_noise.append(es[i])
        xo, p = net.guided_step(xs.cuda(), t, pred, temp, want_probs=True)
        tf.step(i, p.cpu(), p_ref, xo.cpu(), x_ref, es[i], zmax)
        x_last = (xo, x_ref)
    tf.finish('cg_real_guided_trpb')
    # guidance really acted at the read positions (and only there): the guided probabilities differ from the prior
    read0 = sorted(set(res))                                    # positions the value function reads (raw list)
    # final projection (cls_guidance.py:58): frozen positions come back as the parent, the rest as sampled
    proj = algo.project(x_last[0]).cpu()
    ok = tf.last_ok
    want = g['x']
    mutable = torch.zeros(Lq, dtype=torch.bool)
    mutable[[r - 1 for r in res]] = True
    assert torch.equal(proj[:, ~mutable], want[:, ~mutable])
    sel = ok & mutable[None, :]
    assert torch.equal(proj[sel], want[sel])
    assert len(read0) == 15


def test_config1_prior_sampling_b64(env):
    """BASELINE config 1: the sampler of prior_sample.py:73 (no forward_op -> temperature 0) on CreiLOV at the real DiT
    dims, B = 64, 127 steps.  (i) teacher-forced on the first four rows against the reference's own states; (ii) the
    whole batch free-running on the reference's noise: same result twice, legal tokens, strings of the 20 standard
    letters, and agreement with the reference's final tokens (a bf16 denoiser flips near-tie draws, so not all)."""
    from oracle import cls_guidance as ocg
    from oracle import mdlm as omdlm
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    g = load_golden('cg_real_unguided_b64')
    T = int(g['T'])
    B, Lq = g['x'].shape
    es = golden_noise(g, B, Lq)
    net, sd = _mdlm(g, T, Lq)
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq=CREILOV, residues=None)
    algo = Classifier_Guidance_Inpaint(temperature=0., n_max_mutations=None, net=net, data_config=dc)   # prior_sample.py:73
    assert algo.temperature == 0. and algo.forward_op is None
    states = g['states_first'].long()
    n = states.shape[1]
    timesteps = torch.linspace(T - 1, 1, T - 1, dtype=int)
    tf = TeacherForced()
    for i, t in enumerate(timesteps):
        xs = states[i]
        q, zmax = oracle_soft_step(sd, xs, t, T)
        x_ref, p_ref = ocg.cg_step(q, xs, lambda qm: qm, es[i, :n])
        assert torch.equal(x_ref, states[i + 1] if i + 1 < T - 1 else g['x'][:n])
        net.injected_noise.append(es[i, :n])
        xo, p = net.guided_step(xs.cuda(), t, None, 0.0, want_probs=True)
        tf.step(i, p.cpu(), p_ref, xo.cpu(), x_ref, es[i, :n], zmax)
    tf.finish('cg_real_unguided_b64')
    # (ii) free run of all 64 rows through Algo.inference on the same noise
    runs = []
    for _ in range(2):
        for e in es:
            net.injected_noise.append(e)
        np.random.seed(0)
        x, seqs = algo.inference(num_samples=B, detokenize=True)
        runs.append(x.cpu())
        assert len(net.injected_noise) == 0
    assert torch.equal(runs[0], runs[1])
    x = runs[0]
    assert x.shape == (B, Lq) and x.dtype == torch.int64 and int(x.min()) >= 0 and int(x.max()) < 31
    assert len(seqs) == B and all(len(s) == Lq and set(s) <= set(ALPHABET[:20]) for s in seqs)
    agree = float((x == g['x']).float().mean())
    print(f'config 1 free run vs reference final tokens: {agree:.4f} of {x.numel()} positions equal')
    assert agree > 0.9, agree


def test_config5_ensemble_member_guides_batches_of_10(env):
    """BASELINE config 5 (iterativeBO.py:143-237): train a small value-function ensemble on the GPU, draw a member
    (Thompson sampling), sample 10 guided sequences with it.  (i) the 10 rows equal the same global rows of a
    4096-row call bit for bit (small-tile GEMMs and spread jump tiles against the persistent kernels); (ii) a few
    steps teacher-forced against the oracle evaluated with that member's trained weights."""
    from oracle import cls_guidance as ocg
    from oracle import mdlm as omdlm
    from oracle import mlp as omlp
    from models.mdlm import MDLM
    from problem.protein import ProteinPredictor
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    from sgpo_amd import weights as W
    from training.train_classifier import train_classifier_ensemble
    dev, Lq, T, temp, n_members = 'cuda:0', len(CREILOV), 128, 0.064, 4
    sd = W.dit_synthetic_state(seed=0)
    cfg = dict(diffusion='absorbing_state', parameterization='subs', time_conditioning=False,
               model=dict(hidden_size=512, cond_dim=128, n_blocks=8, n_heads=8, length=1024))
    net = MDLM('synthetic', cfg, None, num_steps=T, seq_len=Lq, device=dev, state_dict=sd)
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq=CREILOV, residues=None)
    # a tiny labelled set: variants of the parent, fitness = a fixed random linear score
    rng = np.random.default_rng(0)
    parent = np.array([ALPHABET.index(c) for c in CREILOV])
    X = np.tile(parent, (96, 1))
    pos = rng.integers(0, Lq, (96, 4))
    X[np.arange(96)[:, None], pos] = rng.integers(0, 20, (96, 4))
    wlin = rng.normal(size=(Lq, 20)).astype(np.float32)
    Y = wlin[np.arange(Lq)[None, :], X].sum(1) / 10.0
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.from_numpy(X), torch.from_numpy(Y)),
                                         batch_size=32, shuffle=True)
    clfs = []
    for k in range(n_members):
        torch.manual_seed(42 + k)                              # iterativeBO.py:160: seed + round * n_ensemble + idx
        clfs.append(ProteinPredictor(dc, dict(hidden_dim=64), device=dev))
    before = [c.model.state_dict()['mlp.0.weight'].clone() for c in clfs]
    algo0 = Classifier_Guidance_Inpaint(net, dc, forward_op=clfs[0], temperature=temp, device=dev)
    tc = dict(learning_rate=1e-3, wandb=False, n_epochs=20, n_epochs_no_time=20, batch_size=32)
    train_classifier_ensemble(clfs, net, loader, tc, save_dir=None, project_fn=algo0.project, time_conditioned=True)
    for c, w0 in zip(clfs, before):
        assert not torch.equal(c.model.state_dict()['mlp.0.weight'], w0)        # training moved every member
    np.random.seed(3)
    member = clfs[np.random.randint(0, n_members)]                              # iterativeBO.py:213
    algo = Classifier_Guidance_Inpaint(net, dc, forward_op=member, temperature=temp, device=dev)
    algo.update_model(member)
    # (i) batch of 10 == rows [r0, r0 + 10) of a 4096-row call, short horizon to keep the test quick
    net.timestep = 12
    r0 = 1237
    net.reseed(17, row_offset=0)
    whole = algo.inference(num_samples=4096)
    net.reseed(17, row_offset=r0)
    draw = algo.inference(num_samples=10)
    assert torch.equal(draw, whole[r0:r0 + 10])
    net.reseed(17, row_offset=0)
    assert torch.equal(algo.inference(num_samples=10), whole[:10])
    # (ii) teacher-forced steps with the member's trained weights in the oracle; states: fully masked, then the
    # oracle's own successors
    net.timestep = T
    msd = {k: v.detach().cpu().clone() for k, v in member.model.state_dict().items()}
    opred = omlp.Predictor(msd)
    gen = torch.Generator().manual_seed(5)
    B = 10
    xs = torch.full((B, Lq), 28, dtype=torch.int64)
    tf = TeacherForced()
    for i, t in enumerate(torch.linspace(T - 1, 1, T - 1, dtype=int)[:4]):
        e = torch.empty(B, Lq, 31).exponential_(1, generator=gen)
        q, zmax = oracle_soft_step(sd, xs, t, T)
        x_ref, p_ref = ocg.cg_step(q, xs, lambda qm: omdlm.get_guided_rates(opred, xs, t, qm, 31, temp)[0], e)
        net.injected_noise.append(e)
        xo, p = net.guided_step(xs.cuda(), t, member, temp, want_probs=True)
        tf.step(i, p.cpu(), p_ref, xo.cpu(), x_ref, e, zmax)
        # the trained member's scores through the C ABI equal the oracle's
        torch.testing.assert_close(member.log_likelihood(xs.cuda()).cpu(), opred.log_likelihood(xs), rtol=1e-4, atol=1e-5)
        xs = x_ref
        if i == 1:                                            # re-mask a third so later steps see masked positions too
            xs = xs.clone()
            xs[:, ::3] = 28
    tf.finish('config5_member')


def test_config5_stacked_thompson_draws_equal_one_draw_at_a_time(env):
    """`inference_many`: several Thompson draws (different ensemble members, different Philox keys) stacked into one
    batch -- one denoiser forward per step for all of them -- give every draw exactly the tokens of its own
    `update_model(member); net.reseed(key); inference(10)` call, and `thompson_round(per_rank=...)` consumes them to the
    same result as the one-draw-at-a-time loop (iterativeBO.py:204-237)."""
    import bench
    from problem.protein import ProteinPredictor
    from sgpo_amd import dist as D
    from sgpo_amd import weights as W
    net, pred0, algo, sd, msd = bench.build('cuda:0', 10)
    net.timestep = 24                                   # a short horizon keeps the test quick; the loop is the same
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=119, full_seq=CREILOV, residues=None)
    members = []
    for m in range(4):
        p = ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0')
        p.model.load_state_dict(W.mlp_synthetic_state(seed=200 + m, n_residues=119))
        members.append(p.eval())
    picks, keys = [2, 0, 3, 3, 1], [D.item_seed(9, d) for d in range(5)]
    stacked = algo.inference_many([members[m] for m in picks], keys, num_samples=10)
    for m, key, got in zip(picks, keys, stacked):
        algo.update_model(members[m])
        net.reseed(seed=key, row_offset=0)
        assert torch.equal(got, algo.inference(num_samples=10))
    assert not torch.equal(stacked[2], stacked[3])       # same member, different key

    def run(per_rank):
        rng = np.random.RandomState(5)
        seen = []

        def draw(m, key):
            algo.update_model(members[m])
            net.reseed(seed=key, row_offset=0)
            x = algo.inference(num_samples=10)
            return x, members[m].log_likelihood(x)

        def draw_many(ms, ks):
            xs = algo.inference_many([members[m] for m in ms], ks, num_samples=10)
            return [(x, members[m].log_likelihood(x)) for m, x in zip(ms, xs)]

        def consume(t, s):
            seen.append((t.cpu(), s.cpu()))
            return 4
        out = D.thompson_round(draw, consume, n_needed=14, max_draws=20, pick_member=lambda: int(rng.randint(0, 4)), seed=21,
                               rank=0, world=1, per_rank=per_rank, draw_many=draw_many if per_rank > 1 else None)
        return out, seen
    (a, seen_a), (b, seen_b) = run(1), run(3)
    assert a == b == (16, 4) and len(seen_a) == len(seen_b) == 4
    for (ta, sa), (tb, sb) in zip(seen_a, seen_b):
        assert torch.equal(ta, tb) and torch.equal(sa, sb)
