"""TEST INFRASTRUCTURE (CPU restatement, never on the product path).

Row f-2 of SURVEY.md section 8: what the reference does to a sampled batch between the sampler and the fitness
oracle -- `Algo.project_sequences` (sampling/base.py:40-63) and the filter of
`ProteinPredictorDataset.update_data` (dataset/protein.py:153-183) -- restated on token arrays / hashable
sequences, with the randomness handed in instead of drawn from numpy's global generator.

Pinned by tests/golden/acquire.npz: tools/refgen/make_goldens.py runs the reference's own two functions (numpy's
`random.choice` replaced by a reader of the same noise arrays) and stores their outputs.
"""
import numpy as np


def project_tokens(tokens, full_seq, n_std, u_special=None, n_max_mutations=None, keys=None):
    """sampling/base.py:40-63.  tokens (N, L) ints; returns a repaired copy.

    base.py:45-47: walking the ORIGINAL string, every character outside the 20 letters triggers
    `seq = seq.replace(s, np.random.choice(alphabet))`: the first hit of a character replaces all of its
    occurrences, later hits find nothing left to replace (their draw is wasted).  Here the draw for special
    token k of sequence n is floor(u_special[n, k - n_std] * n_std).
    base.py:50-61 (the caller checks `n_residues > 4 and n_max_mutations is not None`): keep
    min(n_max_mutations, #mutated) of the mutated positions, chosen without replacement -- here the ones with
    the smallest keys[n, j] (ties: smaller j) -- and revert every other position to full_seq."""
    x = np.array(tokens, dtype=np.int64, copy=True)
    N, L = x.shape
    for n in range(N):
        orig = x[n].copy()
        for j in range(L):
            t = int(orig[j])
            if t >= n_std:
                r = min(int(np.float32(u_special[n, t - n_std]) * np.float32(n_std)), n_std - 1)
                x[n][x[n] == t] = r
        if n_max_mutations is not None:
            mutated = [j for j in range(L) if x[n, j] != full_seq[j]]
            k = min(n_max_mutations, len(mutated))
            chosen = set(sorted(mutated, key=lambda j: (float(keys[n, j]), j))[:k])
            for j in range(L):
                if j not in chosen:
                    x[n, j] = full_seq[j]
    return x


class SequenceStore:
    """The bookkeeping of update_data (dataset/protein.py:153-183) on hashable sequences.

    One deliberate difference: the reference's `list(set(new_seqs))` leaves the survivors in the order of the
    interpreter's string hashing (different from run to run); here they stay in batch order, first occurrence
    first.  As sets the results are the same; under truncation to n_needed both keep n_needed admissible
    sequences, not necessarily the same ones."""

    def __init__(self):
        self.seqs = None
        self.round_seqs = None
        self.round = 0
        self.rounds = []

    def update(self, new_seqs, n_needed, BO=False, round=0, unique_only=True):
        """Returns (rows of new_seqs appended, in order; number admissible before the n_needed cut)."""
        if round != self.round:                      # protein.py:160-162
            self.round = round
            self.round_seqs = None
        new_seqs = list(new_seqs)
        idx = list(range(len(new_seqs)))
        if self.seqs is not None:                    # protein.py:167-173
            scope = None
            if BO:
                scope = set(self.seqs)
            elif self.round_seqs is not None and unique_only:
                scope = set(self.round_seqs)
            if scope is not None:
                seen, idx = set(), []
                for i, s in enumerate(new_seqs):
                    if s in seen:
                        continue
                    seen.add(s)
                    if s not in scope:
                        idx.append(i)
        admissible = len(idx)
        if n_needed < admissible:                    # protein.py:178-179
            idx = idx[:max(n_needed, 0)]
        kept = [new_seqs[i] for i in idx]
        self.seqs = kept if self.seqs is None else self.seqs + kept          # protein.py:181-182
        self.round_seqs = kept if self.round_seqs is None else self.round_seqs + kept
        self.rounds += [round] * len(kept)
        return idx, admissible


def landscape_lookup(combo_to_fitness, combos):
    """protein.py:188-197: fitness of each combo, 0 when the data frame lacks it."""
    return np.array([combo_to_fitness.get(c, 0.0) for c in combos], dtype=np.float32)
