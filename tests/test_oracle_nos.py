"""CPU: the NOS oracle vs the golden captured from the reference's MDLM._nos_denoise."""
import pytest
import torch

from conftest import load_golden
from oracle import nos as onos
from sgpo_amd import weights as W

DIM_KEYS = ('hidden', 'cond_dim', 'n_blocks', 'n_heads')


@pytest.mark.parametrize('name', ['nos_small', 'nos_real'])   # nos_real: BASELINE config 4 at full size (real DiT dims, L = 389)
def test_nos_denoise(name):
    g = load_golden(name)
    dims = dict(zip(DIM_KEYS, [int(v) for v in g['dims']]))
    sd = W.dit_synthetic_state(seed=int(g['seed']), vocab=31, **dims)
    for tag in bytes(g['tags'].numpy().tolist()).decode():
        t, nsteps, lr, coef = [float(v) for v in g[f'hp_{tag}']]
        xs, probs, delta = onos.nos_denoise(sd, g['w_cls'], g['b_cls'], g['x'], int(t), int(g['T']), 28, int(nsteps), lr,
                                            coef, g[f'u_{tag}'])
        # fp32 sums in another order than the reference's modules: 1e-4 through 2 blocks, 1.2e-4 observed through 8
        torch.testing.assert_close(probs, g[f'probs_{tag}'], rtol=1e-4 if name == 'nos_small' else 5e-4, atol=1e-7)
        assert torch.equal(xs, g[f'xs_{tag}'])
        assert float(delta.abs().max()) > 0
