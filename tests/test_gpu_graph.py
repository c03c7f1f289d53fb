"""The hot calls of the ABI allocate nothing and never synchronise (include/sgpo_hip.h, "Rules"): a caller may record
them into a HIP graph.  One denoiser forward + one reverse step are captured with torch's stream capture and replayed
on new inputs; the replay must reproduce the eager result bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def rt():
    from sgpo_amd import runtime
    runtime.init('cuda:0')
    return runtime


def test_forward_and_reverse_step_replay_from_a_hip_graph(rt):
    from sgpo_amd import _lib as L, weights as W
    sd = W.dit_synthetic_state(seed=4, vocab=31, hidden=128, cond_dim=32, n_blocks=2, n_heads=2)
    net = rt.DitDenoiser(sd)
    B, Lq, K, mask = 6, 119, 31, 28
    g = torch.Generator().manual_seed(11)

    def draw():
        x = torch.randint(0, 26, (B, Lq), generator=g)
        x[torch.rand(B, Lq, generator=g) < 0.5] = mask
        return x.cuda(), (torch.rand(B, generator=g) * 3).cuda(), torch.rand(B, Lq, K, generator=g).cuda()

    def step(x, sigma, u, logits, x_out):
        net.forward(x, sigma, out=logits)
        rt.reverse_step_mdlm(logits, x, 0.7, 0.4, mask, L.STEP_HARD, noise=u, x_out=x_out)

    # static buffers of the graph
    sx, ss, su = draw()
    s_logits = torch.empty(B, Lq, K, device='cuda')
    s_out = torch.empty_like(sx)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):      # warm-up off the default stream: workspaces, function attributes
        step(sx, ss, su, s_logits, s_out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step(sx, ss, su, s_logits, s_out)
    for _ in range(3):
        x, sigma, u = draw()
        e_logits = torch.empty(B, Lq, K, device='cuda')
        e_out = torch.empty_like(x)
        step(x, sigma, u, e_logits, e_out)
        sx.copy_(x); ss.copy_(sigma); su.copy_(u)
        s_out.fill_(-1)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(s_logits, e_logits)
        assert torch.equal(s_out, e_out)
        assert ((s_out != mask) | (x == mask)).all()


DIM_KEYS = ('hidden', 'cond_dim', 'n_blocks', 'n_heads')
ALPHABET = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'


def _sampler(golden, temp, B_dims=None):
    import sgpo_amd
    sgpo_amd.install_dropin()
    from conftest import load_golden
    from models.mdlm import MDLM
    from problem.protein import ProteinPredictor
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    from sgpo_amd import weights as W
    g = load_golden(golden)
    T = int(g['T'])
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
    algo = Classifier_Guidance_Inpaint(net, dc, forward_op=pred if temp else None, temperature=temp, device='cuda:0')
    return net, pred, algo


@pytest.mark.parametrize('golden,temp,B', [('cg_small_guided', 0.5, 7), ('cg_small_guided', 0.0, 64), ('cg_real_guided', 0.064, 10)])
def test_recorded_sampling_loop_equals_the_eager_loop(rt, golden, temp, B):
    """`net.use_graphs = True`: Classifier_Guidance_Inpaint.inference replays ONE recorded HIP graph (all reverse steps +
    projection) with the Philox key read from device memory.  Tokens equal the eager loop's bit for bit: after a reseed,
    on a second call without reseeding (the call-offset base travels in the key), on a sharded row offset, with
    detokenize; a changed value function re-records.  `cg_real_guided` = the real DiT dims, 127 steps (config 5's
    draw size)."""
    import numpy as np
    net, pred, algo = _sampler(golden, temp)

    def run(graphs, seed, row_offset=0, twice=False, **kw):
        net.use_graphs = graphs
        net.reseed(seed, row_offset=row_offset)
        out = algo.inference(num_samples=B, **kw)
        return (out, algo.inference(num_samples=B, **kw)) if twice else out
    e1, e2 = run(False, 11, twice=True)
    g1, g2 = run(True, 11, twice=True)
    assert torch.equal(e1, g1) and torch.equal(e2, g2) and not torch.equal(g1, g2)
    assert algo._graph_cache is not None
    first = algo._graph_cache[1]
    assert torch.equal(run(False, 12, row_offset=40), run(True, 12, row_offset=40))
    assert algo._graph_cache[1] is first                                    # same graph, new key
    np.random.seed(3)
    xe, se = run(False, 13, detokenize=True)
    np.random.seed(3)
    xg, sg = run(True, 13, detokenize=True)
    assert torch.equal(xe, xg) and se == sg
    if temp:
        with torch.no_grad():
            for p_ in pred.model.parameters():
                p_.mul_(1.5)
        assert torch.equal(run(False, 14), run(True, 14))
        assert algo._graph_cache[1] is not first                            # new weights: recorded again
    net.use_graphs = False


def test_recorded_stacked_thompson_draws_equal_the_eager_call(rt):
    """`inference_many` under `net.use_graphs`: ONE recorded graph serves every wave of stacked draws -- the draws'
    ensemble members are copied into the graph's value-function slots (`sgpo_mlp_copy`), their Philox keys into the key
    buffer.  Two waves with different members and seeds (one draw unguided in both): tokens equal the eager call's."""
    from problem.protein import ProteinPredictor
    from sgpo_amd import weights as W
    net, pred, algo = _sampler('cg_small_guided', 0.5)
    Lq = int(net.length)
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=Lq, full_seq='M' * Lq, residues=None)
    members = []
    for m in range(4):
        p_ = ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0')
        p_.model.load_state_dict(W.mlp_synthetic_state(seed=200 + m, n_residues=Lq))
        members.append(p_.eval())
    waves = [([members[0], members[3], None, members[1]], [5, 6, 7, 8]), ([members[2], members[2], None, members[0]], [9, 10, 11, 2 ** 63 + 5])]
    for ops, seeds in waves:
        net.use_graphs = False
        want = algo.inference_many(ops, seeds, num_samples=6)
        net.use_graphs = True
        got = algo.inference_many(ops, seeds, num_samples=6)
        assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(got, want))
    first = algo._graph_cache_many[1]
    net.use_graphs = True
    algo.inference_many(*waves[0], num_samples=6)
    assert algo._graph_cache_many[1] is first                  # same recording for every wave
    # each draw equals its own one-at-a-time call
    net.use_graphs = False
    algo.update_model(members[3])
    net.reseed(6)
    assert torch.equal(algo.inference(num_samples=6), algo.inference_many(*waves[0], num_samples=6)[1])
