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
