"""HIP-backed mirror of the reference's oracle/inference_oracle.py.

Same call as the reference (`dataset/protein.py:202`):
    _, y = inference_oracle(new_combos, protein=name, model_path=oracle_model_path)
The package is called `fitness`, not `oracle`, because this repository's CPU test oracle already owns
the top-level name `oracle`; INTEGRATION.md shows the one-line import change.

Differences from the reference are limited to where the numbers are computed: the ensemble forward,
the Hamming penalty, max(., 0) and the member mean run in one GPU call.  The checkpoint files under
`model_path` are the user's own `torch.save`d OracleModel state dicts; they are read with
`torch.load(weights_only=True)`.
"""
import os

import torch

AA20 = 'ACDEFGHIKLMNPQRSTVWY'                       # oracle/train_oracle.py:41
PENALTY = {'CreiLOV': (70, 0.99), 'TrpB': (233, 0.99)}   # oracle/inference_oracle.py:49-54

_cache = {}


def hamming_distance(s1, s2):
    return sum(c1 != c2 for c1, c2 in zip(s1, s2))


def penalty(hamming_distances, cutoff=5, rate=0.95):
    if hamming_distances <= cutoff:
        return 1
    return rate ** (hamming_distances - cutoff)


def _read_fasta_first(path):
    seq = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith('>'):
                if seq:
                    break
                continue
            seq.append(line)
    return ''.join(seq)


def _read_fasta_all(path):
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


def _tokens(seqs):
    # a letter outside the alphabet raises ValueError here, as alphabet.index does in the reference
    return torch.tensor([[AA20.index(c) for c in s] for s in seqs], dtype=torch.int64)


def _ensemble(model_path, device):
    from sgpo_amd import runtime as rt
    files = os.listdir(model_path)                     # the reference's member order (inference_oracle.py:31)
    key = (os.path.abspath(model_path), tuple(files), str(device))
    if key not in _cache:
        states = [torch.load(os.path.join(model_path, f), map_location='cpu', weights_only=True) for f in files]
        _cache[key] = rt.FitnessEnsemble(states, device=device)
    return _cache[key]


def inference_oracle(split, protein, model_path, impose_penalty=True, full_seq=None, device=None):
    """Returns (test_loss, predictions float64 (N,)) like oracle/inference_oracle.py:19-72.

    split: list of sequences (or residue combos), or a fasta path.  The 'train'/'validation'/'test'
    splits of data/<protein>/fitness.csv are training-side and not part of this path.
    full_seq: the parent sequence; default: data/<protein>/parent.fasta relative to the working
    directory, as in the reference."""
    if isinstance(split, str):
        if split in ('train', 'validation', 'test'):
            raise NotImplementedError('fitness.csv splits belong to oracle training (out of the sampling path)')
        seqs = _read_fasta_all(split)
    else:
        seqs = list(split)
    if full_seq is None:
        full_seq = _read_fasta_first(f'data/{protein}/parent.fasta')
    full_seq = str(full_seq)
    ens = _ensemble(model_path, device)
    tok = _tokens(seqs)
    if impose_penalty:
        cutoff, rate = PENALTY[protein]                # any other protein: UnboundLocalError in the reference
        # zip(full_seq, seq) stops at the shorter string: a residue combo is compared with the first
        # len(combo) letters of the parent (inference_oracle.py:10-11,56)
        n = tok.shape[1]
        par = [AA20.index(c) if c in AA20 else -1 for c in full_seq[:n]]
        par_t = torch.tensor(par, dtype=torch.int64)
        if len(full_seq) < n:
            raise NotImplementedError('parent shorter than the sequences (never the case in the reference data)')
        y = ens.forward(tok, par_t, cutoff, rate)
    else:
        y = ens.forward(tok, None)
    y = y.cpu()
    n_batches = (len(seqs) + 127) // 128               # len(test_loader), BATCH_SIZE = 128
    test_loss = torch.nn.functional.mse_loss(y, torch.zeros(len(seqs))) / n_batches   # "meaningless" for list splits
    return test_loss, y


def inference_oracle_tokens(tokens, protein, model_path, impose_penalty=True, full_seq=None, device=None):
    """The same ensemble on a token tensor (N, n_res) int64 (ids of the 20 letters = tokenizer ids): the
    acquisition step's fast path (dataset/protein.py:185,202 without the strings).  Returns float64 (N,) on
    the CPU like `inference_oracle`'s second value."""
    ens = _ensemble(model_path, device)
    if impose_penalty:
        if full_seq is None:
            full_seq = _read_fasta_first(f'data/{protein}/parent.fasta')
        cutoff, rate = PENALTY[protein]
        n = tokens.shape[1]
        par_t = torch.tensor([AA20.index(c) if c in AA20 else -1 for c in str(full_seq)[:n]], dtype=torch.int64)
        return ens.forward(tokens, par_t, cutoff, rate).cpu()
    return ens.forward(tokens, None).cpu()
