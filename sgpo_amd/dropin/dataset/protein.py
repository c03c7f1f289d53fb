"""Drop-in `dataset.protein.ProteinPredictorDataset` (reference: dataset/protein.py:10-219; SURVEY.md
section 8 row f-2) with the sequence store, the de-duplication and the fitness evaluation on the GPU.

Same constructor and methods as the reference.  `update_data` also accepts the sampler's token tensor
(N, seq_len) directly, which skips the detokenise -> Python set -> re-tokenise round trip; strings are
tokenised once and take the same path.  Survivors stay in batch order (the reference's `list(set(...))`
order follows the interpreter's string-hash seed); `seqs`, `round_seqs`, `y` and `summary_df` read like
the reference's attributes.
"""
import numpy as np
import torch

from models.base import cfg_get
from sgpo_amd import runtime as rt
from sgpo_amd.tokenizer import MSA_ALPHABET

from .base import DiscreteData


def _read_fasta(path):
    out, cur = [], None
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith('>'):
                cur = []
                out.append(cur)
            elif cur is not None:
                cur.append(line)
    return [''.join(c) for c in out]


class ProteinPredictorDataset(DiscreteData):
    def __init__(self, data_config, tokenizer, n_random_init=0, n_max_mutations=None, from_fasta=None,
                 capacity=1 << 20, max_batch=1 << 16, device=None):
        self.data_config = data_config
        self.name = cfg_get(data_config, 'name')
        data_path = cfg_get(data_config, 'data_path', None)
        self.df = None
        if data_path is not None:
            import pandas as pd
            self.df = pd.read_csv(data_path)
        self.n_max_mutations = n_max_mutations
        self.full_seq_string = cfg_get(data_config, 'full_seq')
        self.seq_len = cfg_get(data_config, 'seq_len')
        self.alphabet_size = cfg_get(data_config, 'alphabet_size')
        residues = cfg_get(data_config, 'residues', None)
        if residues is None:
            self.residues = np.arange(self.seq_len)
        else:
            self.residues = [r - 1 for r in residues]   # zero indexing
        self.n_residues = len(self.residues)
        self.tokenizer = tokenizer
        self.mask = torch.zeros(self.seq_len)
        self.mask[self.residues] = 1
        self.mapping = 'ACDEFGHIKLMNPQRSTVWYBZXJOU'

        self.device = rt.init(device)
        self._set = rt.AcquisitionSet(self.seq_len, capacity, max_batch, device=self.device)
        self._a_to_i = {a: i for i, a in enumerate(MSA_ALPHABET)}
        self._i_to_a = np.array(list(MSA_ALPHABET))
        self._has_seqs = False        # `self.seqs is not None` of the reference
        self._round_start = None      # store row where the current round begins; None = round_seqs is None
        self._table = None            # dense 20^4 landscape (n_residues == 4)
        self.round = 0
        self.y = None

        if from_fasta is not None:
            seqs = _read_fasta(from_fasta)
            self.update_data(seqs, n_needed=len(seqs))
        elif n_random_init > 0:
            if self.n_residues == 4:
                self.df = self.df.sample(n=n_random_init, random_state=cfg_get(data_config, 'seed'))
                self.df['full_sequence'] = self.df['Combo'].apply(self.get_full_sequence)
                seqs = self.df['full_sequence'].values
            else:
                seqs = self.sample_random_variants(n_random_init)
            if len(seqs) < n_random_init:
                print(f'Only found {len(seqs)} unique samples for random initialization')
            self.update_data(seqs, n_needed=n_random_init)

    # ---- host helpers (dataset/protein.py:92-135): same results and the same draws from numpy's global RNG ----
    _LETTERS = np.array(list('ACDEFGHIKLMNPQRSTVWY'))

    def _parent_with(self, sites, letters):
        seq = np.array(list(self.full_seq_string))
        seq[np.asarray(sites, dtype=np.int64)] = letters
        return ''.join(seq)

    def get_full_sequence(self, combo):
        """The parent with the letters of `combo` at the designable residues (landscape rows, protein.py:92-99)."""
        return self._parent_with(self.residues, list(combo))

    def sample_random_variants(self, n_samples):
        """Random variants of the parent for `n_random_init` (protein.py:108-135).  Per variant: when n_max_mutations is
        set, first ONE draw of that many designable residues without replacement; then one letter draw per site, in
        site order."""
        if self.n_max_mutations is not None:
            self.n_max_mutations = min(self.n_max_mutations, self.n_residues)
        out = []
        for _ in range(n_samples):
            sites = self.residues if self.n_max_mutations is None else \
                np.random.choice(self.residues, self.n_max_mutations, replace=False)
            letters = [np.random.choice(self._LETTERS) for _ in sites]
            out.append(self._parent_with(sites, letters))
        return out

    def get_length(self):
        return self.n_residues

    def get_dim(self):
        return self.alphabet_size

    def __len__(self):
        return len(self._set)

    def __getitem__(self, i):
        tok, _ = self._set.read(i, 1)
        return (self._untokenize(tok[0].cpu().numpy()),), self.y[i]

    # ---- reference attributes, materialised on demand ----
    def _untokenize(self, row):
        return ''.join(self._i_to_a[row])

    def _strings(self, first, count):
        if count <= 0:
            return []
        tok, _ = self._set.read(first, count)
        return [self._untokenize(r) for r in tok.cpu().numpy()]

    @property
    def seqs(self):
        return self._strings(0, len(self._set)) if self._has_seqs else None

    @property
    def round_seqs(self):
        if self._round_start is None:
            return None
        return self._strings(self._round_start, len(self._set) - self._round_start)

    @property
    def tokens(self):
        """(n, seq_len) int64 device tensor of every stored sequence, in append order."""
        return self._set.read(0, len(self._set))[0]

    @property
    def summary_df(self):
        if not self._has_seqs:
            return None
        import pandas as pd
        _, rounds = self._set.read(0, len(self._set))
        return pd.DataFrame({'sequence': self.seqs, 'round': rounds.cpu().numpy(), 'fitness': self.y.numpy()})

    def save_data(self, save_path):
        self.summary_df.to_csv(save_path, index=False)

    # ---- the acquisition step (dataset/protein.py:153-213) ----
    def _tokenize(self, new_seqs):
        if torch.is_tensor(new_seqs):
            return new_seqs.to(self.device, torch.int64).contiguous()
        new_seqs = list(new_seqs)
        if not new_seqs:
            return torch.empty(0, self.seq_len, dtype=torch.int64, device=self.device)
        arr = np.array([[self._a_to_i[c] for c in s] for s in new_seqs], dtype=np.int64)
        return torch.from_numpy(arr).to(self.device)

    def _fitness(self, tok):
        if tok.shape[0] == 0:
            return torch.zeros(0)
        res = torch.as_tensor(np.asarray(self.residues), dtype=torch.int64, device=self.device)
        if self.n_residues == 4:
            if self._table is None:
                table = torch.zeros(20 ** 4, dtype=torch.float32)
                idx = [((self._a_to_i[c[0]] * 20 + self._a_to_i[c[1]]) * 20 + self._a_to_i[c[2]]) * 20 + self._a_to_i[c[3]]
                       for c in self.df['Combo'].values]
                table[torch.tensor(idx, dtype=torch.int64)] = torch.tensor(self.df['fitness'].values, dtype=torch.float32)
                self._table = table.to(self.device)
            return rt.landscape_lookup(self._table, tok, np.asarray(self.residues, dtype=np.int32)).cpu()
        from fitness.inference_oracle import inference_oracle_tokens
        return inference_oracle_tokens(tok[:, res], protein=cfg_get(self.data_config, 'name'),
                                       model_path=cfg_get(self.data_config, 'oracle_model_path'), device=self.device).float()

    def update_data(self, new_seqs, n_needed, BO=False, round=0, unique_only=True):
        """new_seqs: list of strings (as in the reference) or the sampler's (N, seq_len) token tensor.
        Returns the number of sequences appended."""
        if round != self.round:
            self.round = round
            self._round_start = None
        tok = self._tokenize(new_seqs)
        scope = rt.AcquisitionSet.KEEP_ALL
        if self._has_seqs:
            if BO:
                scope = rt.AcquisitionSet.NOT_IN_STORE
            elif self._round_start is not None and unique_only:
                scope = rt.AcquisitionSet.NOT_IN_ROUND
        before = len(self._set)
        kept, _ = self._set.update(tok, n_needed, scope, round)
        self._has_seqs = True
        if self._round_start is None:
            self._round_start = before
        y_new = self._fitness(tok[kept])
        self.y = y_new if self.y is None else torch.cat([self.y, y_new], dim=0)
        return int(kept.numel())
