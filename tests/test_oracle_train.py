"""Row f-3 on the CPU: the restatement of train_classifier's loop (oracle/train_mlp.py) against the weights the
reference's own train_classifier produced on the same batches (tests/golden/train_mlp.npz)."""
import torch

from conftest import load_golden
from oracle import train_mlp
from sgpo_amd import weights as W

ORDER = ['time_embedder.0.W', 'time_embedder.1.weight', 'time_embedder.1.bias', 'embedder.weight', 'embedder.bias',
         'mlp.0.weight', 'mlp.0.bias', 'mlp.2.weight', 'mlp.2.bias', 'mlp.4.weight', 'mlp.4.bias', 'cls_head.0.weight',
         'cls_head.0.bias', 'cls_head.2.weight', 'cls_head.2.bias']


def case(g, name):
    nb = g[f'{name}_nb'].tolist()
    batches = [(g[f'{name}_xt'][i, :n], g[f'{name}_t'][i, :n], g[f'{name}_y'][i, :n]) for i, n in enumerate(nb)]
    residues = g[f'{name}_residues'].tolist() or None
    sd0 = W.mlp_synthetic_state(seed=int(g[f'{name}_seed']), n_residues=int(g[f'{name}_n_res']))
    sd1 = {k: g[f'{name}_w1_{k}'] for k in ORDER}
    return sd0, batches, residues, sd1, g[f'{name}_losses']


def test_training_restatement_matches_reference_weights():
    g = load_golden('train_mlp')
    for name in ('full', 'sites', 'clean'):
        sd0, batches, residues, sd1, losses = case(g, name)
        mine, my_losses = train_mlp.train(sd0, batches, residues=residues)
        for k in ORDER:
            torch.testing.assert_close(mine[k], sd1[k], rtol=0, atol=1e-7)
        torch.testing.assert_close(torch.tensor(my_losses), losses, rtol=1e-6, atol=0)
        assert torch.equal(mine['time_embedder.0.W'], sd0['time_embedder.0.W'])      # never trained
        assert not torch.equal(mine['embedder.weight'], sd0['embedder.weight'])
