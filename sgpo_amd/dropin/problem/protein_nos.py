"""Value function of the NOS path: the head of `DITClassifier` (reference: dit.py:420-515) as it is
used by `_nos_denoise` -- mean-pool the (frozen backbone's) last hidden state, Linear(hidden, 1).
The reference's NOS config points at module paths that do not exist (defect D5); this class gives
the sampler the one piece it reads: `output_layer`."""
import torch
import torch.nn as nn


class NOSHead(nn.Module):
    def __init__(self, hidden_size=512):
        super().__init__()
        self.pooling = 'mean'
        self.output_layer = nn.Linear(hidden_size, 1)

    def forward(self, indices_or_one_hots=None, sigma=None, x_emb=None, attention_mask=None):
        if x_emb is None:
            raise NotImplementedError('NOSHead scores hidden states (x_emb=...); run the backbone first')
        return self.output_layer(x_emb.mean(dim=1))
