"""Mirror of `sampling.base.Algo` (reference: sampling/base.py:5-72): residue mask,
`full_seq` tokenisation, and the host-side string repair `project_sequences`
(numpy RNG, defect D4 semantics kept: every occurrence of a bad character is replaced)."""
from abc import ABC, abstractmethod

import numpy as np
import torch

from models.base import cfg_get


class Algo(ABC):
    N_STANDARD = 20     # the standard amino acids lead the alphabet; whatever follows is a special character

    def __init__(self, net, forward_op=None, data_config=None, n_max_mutations=None, device='cuda', causal_LM=False):
        self.net, self.forward_op, self.data_config = net, forward_op, data_config
        self.n_max_mutations = n_max_mutations
        self.device = getattr(net, '_dev', None) or device      # the rank's own GPU when the net is HIP-backed

        letters = cfg_get(data_config, 'alphabet')
        self.alphabet, self.special_tokens = list(letters[:self.N_STANDARD]), list(letters[self.N_STANDARD:])
        self.seq_len = cfg_get(data_config, 'seq_len')
        self.full_seq_string = cfg_get(data_config, 'full_seq')
        self.all_residues = list(range(self.seq_len))
        listed = cfg_get(data_config, 'residues')               # 1-indexed in the yaml (configs/data/TrpB.yaml:6)
        self.residues = self.all_residues[:] if listed is None else [int(r) - 1 for r in listed]
        self.n_residues = len(self.residues)

        designable = torch.zeros(self.seq_len, dtype=torch.int32)
        designable[self.residues] = 1
        self.mask = designable.to(self.device)
        if causal_LM:
            self.full_seq = self.full_seq_string
        else:
            ids = np.concatenate([np.atleast_1d(net.tokenizer.tokenize(ch)) for ch in self.full_seq_string])
            self.full_seq = torch.from_numpy(ids).to(self.device).int()

    # ---- string repair of the generations (sampling/base.py:40-63), numpy's global RNG as there -------------------
    def _replace_specials(self, seq):
        """One np.random.choice per non-standard character met while walking the ORIGINAL string (a repeated character
        draws again, although the first draw already replaced every occurrence: defect D4's accounting, kept because
        the draws shift everything after them in numpy's stream)."""
        chosen = {}
        for ch in seq:
            if ch not in self.alphabet:
                draw = np.random.choice(self.alphabet)
                chosen.setdefault(ch, draw)
        return seq.translate(str.maketrans(chosen)) if chosen else seq

    def _limit_mutations(self, seq):
        """Keep a random subset of at most n_max_mutations of the positions that differ from the parent."""
        child = np.frombuffer(seq.encode('ascii'), dtype=np.uint8)
        parent = np.frombuffer(self.full_seq_string.encode('ascii'), dtype=np.uint8)
        differs = np.flatnonzero(child != parent)
        keep = np.random.choice(differs, min(self.n_max_mutations, differs.size), replace=False)
        out = parent.copy()
        out[keep] = child[keep]
        return out.tobytes().decode('ascii')

    def project_sequences(self, sequences):
        repaired = [self._replace_specials(s) for s in sequences]          # all step-1 draws come first
        if self.n_residues > 4 and self.n_max_mutations is not None:
            repaired = [self._limit_mutations(s) for s in repaired]
        return repaired

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
