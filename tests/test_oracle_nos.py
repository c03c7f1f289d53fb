"""CPU: the NOS oracle vs the golden captured from the reference's MDLM._nos_denoise."""
import torch

from conftest import load_golden
from oracle import nos as onos
from sgpo_amd import weights as W

DIM_KEYS = ('hidden', 'cond_dim', 'n_blocks', 'n_heads')


def test_nos_denoise():
    g = load_golden('nos_small')
    dims = dict(zip(DIM_KEYS, [int(v) for v in g['dims']]))
    sd = W.dit_synthetic_state(seed=int(g['seed']), vocab=31, **dims)
    for tag in 'abc':
        t, nsteps, lr, coef = [float(v) for v in g[f'hp_{tag}']]
        xs, probs, delta = onos.nos_denoise(sd, g['w_cls'], g['b_cls'], g['x'], int(t), int(g['T']), 28, int(nsteps), lr,
                                            coef, g[f'u_{tag}'])
        torch.testing.assert_close(probs, g[f'probs_{tag}'], rtol=1e-4, atol=1e-7)
        assert torch.equal(xs, g[f'xs_{tag}'])
        assert float(delta.abs().max()) > 0
