"""ORACLE (test infrastructure, not product): CPU fp32 restatement of the MLP
value function.

Follows /root/reference/problem/protein.py:
  GaussianFourierProjection :43-56
  MLPModel.forward          :97-121  (incl. defect D3: `seq[:, residues]` uses the
                                      1-indexed config list unchanged)
  ProteinPredictor.__call__ / log_likelihood :30-38
Pinned by tests/golden/mlp_*.npz.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def time_embed(sd, t):
    W = sd['time_embedder.0.W']
    x_proj = t[:, None] * W[None, :] * 2 * np.pi
    ff = torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)
    return F.linear(ff, sd['time_embedder.1.weight'], sd['time_embedder.1.bias'])


def mlp_forward(sd, seq, t, residues=None, alphabet_size=31, time_conditioned=True):
    """MLPModel.forward: seq (B,D) int64, t (B,) -> (B,1)."""
    if residues is not None:
        seq = seq[:, np.array(residues)]
    oh = F.one_hot(seq.long(), num_classes=alphabet_size).float()
    feat = F.linear(oh, sd['embedder.weight'], sd['embedder.bias'])
    feat = feat.view(feat.size(0), -1)
    if time_conditioned:
        feat = torch.cat([feat, time_embed(sd, t.float())], dim=-1)
    h = F.relu(F.linear(feat, sd['mlp.0.weight'], sd['mlp.0.bias']))
    h = F.relu(F.linear(h, sd['mlp.2.weight'], sd['mlp.2.bias']))
    h = F.linear(h, sd['mlp.4.weight'], sd['mlp.4.bias'])
    h = F.relu(F.linear(h, sd['cls_head.0.weight'], sd['cls_head.0.bias']))
    return F.linear(h, sd['cls_head.2.weight'], sd['cls_head.2.bias'])


class Predictor:
    """ProteinPredictor protocol: callable(x, t) -> (B,1); log_likelihood(x) -> (B,)."""

    def __init__(self, sd, residues=None, alphabet_size=31, time_conditioned=True):
        self.sd, self.residues = sd, residues
        self.alphabet_size, self.time_conditioned = alphabet_size, time_conditioned

    def __call__(self, x, t):
        return mlp_forward(self.sd, x, t, self.residues, self.alphabet_size, self.time_conditioned)

    def log_likelihood(self, x, y=None):
        return self(x, torch.zeros(x.shape[0])).squeeze(-1)
