"""Drop-in `problem.protein` (reference: problem/protein.py:9-121).

`ProteinPredictor` / `MLPModel` stay `nn.Module`s with the reference's attribute names
(`embedder`, `time_embedder`, `mlp`, `cls_head`), so `train_classifier` (PyTorch autograd,
out of the hot path) and `torch.save/load` of whole predictors (iterativeBO.py:178) keep
working.  For sampling, `hip_value_function()` hands the current weights to the library:
the fused jump-ratio kernel and the HIP forward take over (eval mode, no grad).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from models.base import cfg_get


class GaussianFourierProjection(nn.Module):
    def __init__(self, embed_dim, scale=30.0):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embed_dim // 2) * scale, requires_grad=False)

    def forward(self, x):
        x_proj = x[:, None] * self.W[None, :] * 2 * np.pi
        return torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)


class MLPModel(nn.Module):
    def __init__(self, data_config, model_config, time_conditioned=True):
        super().__init__()
        self.alphabet_size = cfg_get(data_config, 'alphabet_size')
        residues = cfg_get(data_config, 'residues')
        if residues is not None:
            self.residues = np.array(list(residues))
            self.n_residues = len(self.residues)
        else:
            self.residues = None
            self.n_residues = len(cfg_get(data_config, 'full_seq'))
        self.hidden_dim = cfg_get(model_config, 'hidden_dim')
        self.time_conditioned = time_conditioned
        self.time_embedder = nn.Sequential(GaussianFourierProjection(embed_dim=self.hidden_dim),
                                           nn.Linear(self.hidden_dim, self.hidden_dim))
        self.embedder = nn.Linear(self.alphabet_size, self.hidden_dim)
        input_dim = self.hidden_dim * (self.n_residues + 1) if time_conditioned else self.hidden_dim * self.n_residues
        self.mlp = nn.Sequential(nn.Linear(input_dim, self.hidden_dim), nn.ReLU(),
                                 nn.Linear(self.hidden_dim, self.hidden_dim), nn.ReLU(),
                                 nn.Linear(self.hidden_dim, self.hidden_dim))
        self.cls_head = nn.Sequential(nn.Linear(self.hidden_dim, self.hidden_dim), nn.ReLU(),
                                      nn.Linear(self.hidden_dim, 1))

    def read_positions(self):
        """Positions of the input row the model reads: the raw config list (defect D3)."""
        return [int(r) for r in self.residues] if self.residues is not None else list(range(self.n_residues))

    def forward(self, seq=None, t=None):
        """Autograd-capable torch forward: used for training (out of the sampling hot path)."""
        if self.residues is not None:
            seq = seq[:, self.residues]
        if seq.dim() == 3:
            # rows that are already one-hot (B, D, S <= alphabet): the Taylor-guidance branch differentiates through them
            # (models/mdlm.py:377-394; the reference one-hots them a second time here and fails)
            seq_encoded = F.pad(seq.float(), (0, self.alphabet_size - seq.shape[-1]))
        else:
            seq_encoded = F.one_hot(seq.long(), num_classes=self.alphabet_size).float()
        feat = self.embedder(seq_encoded)
        if self.time_conditioned:
            time_embed = self.time_embedder(t)
            feat = torch.cat([feat.view(feat.size(0), -1), time_embed], dim=-1)
        feat = self.mlp(feat)
        return self.cls_head(feat)


class ProteinPredictor(nn.Module):
    def __init__(self, data_config, model_config, checkpoint=None, device='cuda'):
        super().__init__()
        self.device = device
        self.model = MLPModel(data_config, model_config)
        self.loss_fn = nn.CrossEntropyLoss()
        if checkpoint and checkpoint != 'None':
            self.model.load_state_dict(torch.load(checkpoint, weights_only=True))
        self.model.to(device)
        self._hip = None
        self._hip_key = None

    def _weights_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.model.parameters())

    def hip_value_function(self):
        """`sgpo_amd.runtime.MlpValueFunction` built from the CURRENT weights (rebuilt after training)."""
        from sgpo_amd import runtime as rt
        key = self._weights_key()
        if self._hip is None or self._hip_key != key:
            sd = {k: v.detach() for k, v in self.model.state_dict().items()}
            self._hip = rt.MlpValueFunction(sd, self.model.read_positions(), alphabet=self.model.alphabet_size,
                                            hidden=self.model.hidden_dim,
                                            time_conditioned=self.model.time_conditioned,
                                            device=next(self.model.parameters()).device)
            self._hip_key = key
        return self._hip

    def __getstate__(self):
        st = self.__dict__.copy()
        st['_hip'] = None
        st['_hip_key'] = None
        return st

    def __call__(self, inputs, t):
        """inputs (B, D), t (B) -> (B, 1)."""
        if not torch.is_grad_enabled() and not self.model.training and inputs.is_cuda and inputs.dim() == 2:
            tt = torch.as_tensor(t, dtype=torch.float32, device=inputs.device)
            return self.hip_value_function().forward(inputs, tt).unsqueeze(-1)
        return self.model(inputs, t)

    def log_likelihood(self, inputs, y=None):
        return self(inputs, torch.zeros(inputs.shape[0], device=inputs.device)).squeeze(-1)


# names of the reference module that are outside the sampling hot path -- Embedder, SampledDeterministicModel (training/train_GP_classifier.py:11,14) --
# resolve to the reference's own definitions when its code base is on sys.path
from sgpo_amd.dropin._fallthrough import forward_missing_names  # noqa: E402

__getattr__ = forward_missing_names('problem', 'protein')
