"""Restated `evodiff.utils.Tokenizer(sequences=True)` (third party, not vendored).

Pinned by the reference only through its config mirror of the alphabet
(configs/data/CreiLOV.yaml:10, configs/data/TrpB.yaml:10) and its call sites:
`tokenize(s)` indexes `s[0]` (sampling/base.py:24,38; collaters.py:554-560),
`K = 26` (models/d3pm.py:16,23-24), mask `#`=28, pad `!`=30.
`q_random_schedule` is restated from the published EvoDiff algorithm
 -- parity unpinned.
"""
import numpy as np
import torch

MSA_ALPHABET = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'


class Tokenizer:
    def __init__(self, sequences=True, **_):
        self.alphabet = list(MSA_ALPHABET)
        self.a_to_i = {a: i for i, a in enumerate(self.alphabet)}
        self.i_to_a = np.array(self.alphabet)
        self.sequences = sequences
        self.K = 26 if sequences else len(self.alphabet)
        self.gap_id = self.a_to_i['-']
        self.stop_id = self.a_to_i['*']
        self.mask_id = self.a_to_i['#']
        self.start_id = self.a_to_i['@']
        self.pad_id = self.a_to_i['!']

    def tokenize(self, seq):
        return np.array([self.a_to_i[a] for a in seq[0]])

    def tokenize_string(self, s):
        return np.array([self.a_to_i[a] for a in s], dtype=np.int64)

    def untokenize(self, x):
        if torch.is_tensor(x):
            x = x.detach().cpu().numpy()
        return ''.join(self.i_to_a[np.asarray(x, dtype=np.int64)])

    def q_random_schedule(self, timesteps=500, accumulate='fp64'):
        """Uniform-transition schedule beta_t = 1/(T - t + 1); returns (Q_bar, Q) fp32 (T,K,K).

        `accumulate='fp64'` forms the cumulative product in fp64 and casts at the
        end (recalled upstream behaviour: sigma[T-1] = +inf); 'fp32' accumulates in
        fp32 (sigma[T-1] ~ 17.98).  See SURVEY.md section 7 "hard parts".
        """
        K = self.K
        dt = torch.float64 if accumulate == 'fp64' else torch.float32
        eye = torch.eye(K, dtype=dt)
        prod = eye.clone()
        qs, qbars = [], []
        for t in range(1, timesteps + 1):
            beta = 1.0 / (timesteps - t + 1)
            q = torch.full((K, K), beta / K, dtype=dt) + eye * (1.0 - beta)
            prod = prod @ q
            qs.append(q.float())
            qbars.append(prod.float())
        return torch.stack(qbars), torch.stack(qs)
