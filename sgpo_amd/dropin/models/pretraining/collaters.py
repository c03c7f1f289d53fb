"""Drop-in `models.pretraining.collaters` for the sampling-side drivers (reference:
models/pretraining/collaters.py:542-589): `collate_fn_mapping['cls_guidance']` is what pareto.py:10 and
iterativeBO.py:13 import to batch `ProteinPredictorDataset` items for value-function training.  The DPO /
pre-training collaters belong to out-of-scope components and raise on use."""
import numpy as np
import torch


class PredictorCollater(object):
    """Tokenises [(seq_str,), ...] items into (x (B, L) int64, y (B,) float32)."""

    def __init__(self, tokenizer):
        self.tokenizer = tokenizer

    def encode_seqs(self, seqs):
        a_to_i = getattr(self.tokenizer, 'a_to_i', None)
        if a_to_i is not None:      # one table lookup per character instead of a tokenize() call per character
            return torch.from_numpy(np.array([[a_to_i[c] for c in seq] for seq in seqs], dtype=np.int64))
        x = []
        for seq in seqs:
            data = [self.tokenizer.tokenize(s) for s in seq]
            x.append(torch.from_numpy(np.array(data)))
        return torch.stack(x).squeeze(-1)

    def __call__(self, batch):
        seqs, y = zip(*batch)
        seqs = [seq[0] for seq in seqs]
        x = self.encode_seqs(seqs)
        y = torch.tensor(y, dtype=torch.float32)
        return x, y


class CausalCollater(object):
    def __init__(self, *a, **k):
        raise NotImplementedError('DPO / causal-LM fine-tuning is outside the sampling path (SURVEY.md section 2 row 21)')


collate_fn_mapping = {'cls_guidance': PredictorCollater, 'DPO': CausalCollater}
