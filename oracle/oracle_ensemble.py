"""CPU restatement of the fitness-oracle ensemble (TEST INFRASTRUCTURE: only tests/, smoke() and
bench.py's cpu_baseline may import this).

Follows /root/reference/oracle/inference_oracle.py:10-72 and oracle/train_oracle.py:39-74:
  * one-hot over the 20-letter alphabet 'ACDEFGHIKLMNPQRSTVWY', flattened position-major
    (train_oracle.py:39-48); a letter outside it raises ValueError (`str.index`);
  * per member: fc2(relu(fc1(x))) in fp32, eval mode (dropout off) (train_oracle.py:61-74);
  * penalty rate**(hd - cutoff) for hd > cutoff, hd = Hamming distance to the parent over
    `zip(parent, seq)` -- i.e. over the first len(seq) letters of the parent when the sequence is a
    residue combo (inference_oracle.py:10-17,47-58; CreiLOV cutoff 70, TrpB 233, rate 0.99);
    the product is formed in float64 (fp32 prediction x python float);
  * max(., 0) per member, mean over members in float64 (inference_oracle.py:62-68).
"""
import numpy as np
import torch

AA20 = 'ACDEFGHIKLMNPQRSTVWY'
PENALTY = {'CreiLOV': (70, 0.99), 'TrpB': (233, 0.99)}


def onehot(seqs):
    X = np.zeros((len(seqs), len(seqs[0]) * 20), dtype=np.float32)
    for n, s in enumerate(seqs):
        for i, aa in enumerate(s):
            X[n, i * 20 + AA20.index(aa)] = 1.0
    return torch.from_numpy(X)


def penalty(hd, cutoff, rate):
    return 1 if hd <= cutoff else rate ** (hd - cutoff)


def ensemble_fitness(states, seqs, parent, protein, impose_penalty=True):
    """states: list of OracleModel state dicts; seqs: list[str]; returns float64 tensor (len(seqs),)."""
    X = onehot(seqs)
    allp = np.zeros((len(states), len(seqs)))
    for m, sd in enumerate(states):
        h = torch.relu(torch.nn.functional.linear(X, sd['fc1.weight'], sd['fc1.bias']))
        pred = torch.nn.functional.linear(h, sd['fc2.weight'], sd['fc2.bias']).numpy().reshape(-1)
        if impose_penalty:
            cutoff, rate = PENALTY[protein]
            hd = [sum(c1 != c2 for c1, c2 in zip(parent, s)) for s in seqs]
            pred = pred * np.array([penalty(d, cutoff, rate) for d in hd])
        allp[m] = np.maximum(pred, 0)
    return torch.tensor(np.mean(allp, axis=0))
