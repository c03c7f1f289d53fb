"""TEST INFRASTRUCTURE (CPU restatement, never on the product path).

Row f-3 of SURVEY.md section 8: training/train_classifier.py:7-64 -- `train_step` (zero_grad, forward of the
classifier on the noised and projected batch, MSELoss, backward, Adam step) -- restated on a state dict with the
batches, timesteps and noised tokens handed in (the reference draws them from torch's global generators).
Pinned by tests/golden/train_mlp.npz: the reference's own train_classifier on the same batches.
"""
import torch

from . import mlp as omlp


def train(sd, batches, residues=None, lr=1e-3, alphabet_size=31, time_conditioned=True):
    """sd: MLPModel state dict (cloned); batches: list of (xt (B, L) int64, t (B,) int64/float, y (B,)).
    Returns (trained state dict, list of per-step losses)."""
    params = {k: v.clone().float().requires_grad_(k != 'time_embedder.0.W') for k, v in sd.items()}
    opt = torch.optim.Adam([p for p in params.values() if p.requires_grad], lr=lr)       # train_classifier.py:23
    crit = torch.nn.MSELoss()                                                             # train_classifier.py:24
    losses = []
    for xt, t, y in batches:
        opt.zero_grad()                                                                   # train_classifier.py:12
        y_pred = omlp.mlp_forward(params, xt, t, residues, alphabet_size, time_conditioned).squeeze()   # :15
        loss = crit(y_pred, y.float())                                                    # :16
        loss.backward()                                                                   # :17
        opt.step()                                                                        # :18
        losses.append(loss.item())
    return {k: v.detach() for k, v in params.items()}, losses
