"""Mirror of `sampling.base.Algo` (reference: sampling/base.py:5-72): residue mask,
`full_seq` tokenisation, and the host-side string repair `project_sequences`
(numpy RNG, defect D4 semantics kept: every occurrence of a bad character is replaced)."""
from abc import ABC, abstractmethod

import numpy as np
import torch

from models.base import cfg_get


class Algo(ABC):
    def __init__(self, net, forward_op=None, data_config=None, n_max_mutations=None, device='cuda', causal_LM=False):
        self.net = net
        self.forward_op = forward_op
        self.data_config = data_config
        self.n_max_mutations = n_max_mutations
        dev = getattr(net, '_dev', None)
        self.device = dev if dev is not None else device

        alphabet = cfg_get(data_config, 'alphabet')
        self.alphabet = list(alphabet[:20])  # only the 20 standard amino acids
        self.special_tokens = list(alphabet[20:])

        self.seq_len = cfg_get(data_config, 'seq_len')
        full_seq = cfg_get(data_config, 'full_seq')
        self.full_seq = [self.net.tokenizer.tokenize(s) for s in full_seq] if not causal_LM else full_seq
        self.full_seq_string = full_seq

        residues = cfg_get(data_config, 'residues')
        if residues is None:
            self.residues = list(range(self.seq_len))
        else:
            self.residues = [i - 1 for i in residues]  # convert to 0-indexing
        self.all_residues = list(range(self.seq_len))

        self.n_residues = len(self.residues)
        self.mask = torch.zeros(self.seq_len)
        self.mask[self.residues] = 1
        self.mask = self.mask.to(self.device).int()
        if not causal_LM:
            self.full_seq = torch.from_numpy(np.array(self.full_seq)).to(self.device).int().squeeze(-1)

    def project_sequences(self, sequences):
        new_sequences = []
        for seq in sequences:
            for s in seq:
                if s not in self.alphabet:
                    seq = seq.replace(s, np.random.choice(self.alphabet))
            new_sequences.append(seq)

        if self.n_residues > 4 and self.n_max_mutations is not None:
            for i, seq in enumerate(new_sequences):
                mutated = [j for j, (r1, r2) in enumerate(zip(seq, self.full_seq_string)) if r1 != r2]
                max_mutations = min(self.n_max_mutations, len(mutated))
                mutated = np.random.choice(mutated, max_mutations, replace=False)
                seq = list(seq)
                for j, r in enumerate(seq):
                    if j not in mutated:
                        seq[j] = self.full_seq_string[j]
                new_sequences[i] = ''.join(seq)
        return new_sequences

    def project_tokens(self, x, generator=None):
        """`project_sequences` on the token tensor (N, L) itself (HIP kernel):
        no detokenise / re-tokenise round trip.  Same distribution as the string version; the draws come from
        `generator` (torch) instead of numpy's global state.  Returns a repaired copy."""
        from sgpo_amd import runtime as rt
        x = x.to(torch.int64).contiguous().clone()
        n_special = len(self.special_tokens)
        u = torch.rand(x.shape[0], n_special, device=x.device, generator=generator)
        cut = self.n_max_mutations if (self.n_residues > 4 and self.n_max_mutations is not None) else None
        keys = torch.rand(x.shape, device=x.device, generator=generator) if cut is not None else None
        return rt.project_tokens(x, self.full_seq.to(torch.int64), len(self.alphabet), n_special, u, cut, keys)

    def _untokenize_batch(self, x):
        xs = x.detach().cpu()
        return [self.net.tokenizer.untokenize(s) for s in xs]

    @abstractmethod
    def inference(self, observation=None, num_samples=1, **kwargs):
        pass
