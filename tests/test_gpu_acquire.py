"""GPU parity of row f-2 through the C ABI: token-level project_sequences and the
acquisition set behind update_data, against the outputs of the reference's own functions
(tests/golden/acquire_project.npz, acquire_update.json), the CPU restatement on larger random batches, and
size-independent properties at the headline batch size."""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import acquisition as acq

pytestmark = pytest.mark.gpu
ALPHABET = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'


@pytest.fixture(scope='module')
def rt():
    import sgpo_amd
    sgpo_amd.install_dropin()
    from sgpo_amd import runtime
    runtime.init('cuda:0')
    return runtime


def _tok(seqs):
    return torch.tensor([[ALPHABET.index(c) for c in s] for s in seqs], dtype=torch.int64).reshape(len(seqs), -1)


def test_project_tokens_bit_exact_against_reference_outputs(rt):
    g = np.load(os.path.join(GOLDEN, 'acquire_project.npz'))
    for case in 'abcd':
        n_max = int(g[f'{case}_n_max'])
        x = torch.from_numpy(g[f'{case}_x']).cuda()
        y = rt.project_tokens(x.clone(), torch.from_numpy(g[f'{case}_full']), 20, 11, torch.from_numpy(g[f'{case}_u_special']),
                              None if n_max < 0 else n_max, torch.from_numpy(g[f'{case}_keys']))
        assert torch.equal(y.cpu(), torch.from_numpy(g[f'{case}_y'])), case


def test_project_tokens_large_batch_against_restatement(rt):
    rng = np.random.default_rng(0)
    for L, N, n_max in ((119, 4096, 10), (389, 512, 25), (1024, 64, 300), (7, 33, 0)):
        full = rng.integers(0, 20, L)
        x = np.tile(full, (N, 1))
        m = rng.random((N, L)) < 0.2
        x[m] = rng.integers(0, 31, int(m.sum()))
        u = rng.random((N, 11)).astype(np.float32)
        k = rng.random((N, L)).astype(np.float32)
        k[:, ::3] = 0.25                                  # plenty of ties
        want = acq.project_tokens(x[:256], full, 20, u, n_max, k)
        got = rt.project_tokens(torch.from_numpy(x).cuda(), torch.from_numpy(full), 20, 11, torch.from_numpy(u), n_max,
                                torch.from_numpy(k)).cpu().numpy()
        assert np.array_equal(got[:256], want), (L, N)
        assert (got < 20).all() and ((got != full).sum(axis=1) <= n_max).all()
        # survivors of the cut are mutations the input had
        changed = got != full
        assert ((got == np.where(x < 20, x, got)) | ~changed).all()


def test_project_tokens_rejects_bad_input(rt):
    x = torch.zeros(4, 16, dtype=torch.int64, device='cuda')
    x[2, 3] = 31
    with pytest.raises(ValueError):
        rt.project_tokens(x, None, 20, 11, torch.rand(4, 11))
    x[2, 3] = 25
    with pytest.raises(ValueError):
        rt.project_tokens(x.clone(), None, 20, 11, None)
    y = rt.project_tokens(x.clone(), None, 20, 11, torch.full((4, 11), 0.999999))
    assert int(y[2, 3]) == 19


def _run_script(rt, calls, L, capacity=4096, max_batch=256):
    dev_set = rt.AcquisitionSet(L, capacity=capacity, max_batch=max_batch)
    store = acq.SequenceStore()
    has_seqs, round_start, cur_round = False, None, 0
    for c in calls:
        if c['round'] != cur_round:
            cur_round, round_start = c['round'], None
        scope = 0
        if has_seqs:
            if c['BO']:
                scope = 1
            elif round_start is not None and c['unique_only']:
                scope = 2
        before = len(dev_set)
        tok = _tok(c['batch']).cuda() if c['batch'] else torch.empty(0, L, dtype=torch.int64, device='cuda')
        kept, adm = dev_set.update(tok, c['n_needed'], scope, c['round'])
        has_seqs = True
        if round_start is None:
            round_start = before
        idx, adm_ref = store.update(c['batch'], c['n_needed'], BO=c['BO'], round=c['round'], unique_only=c['unique_only'])
        assert kept.cpu().tolist() == idx and adm == adm_ref
        if not c['truncated'] and not c.get('diverged'):
            assert sorted(c['batch'][i] for i in idx) == c['ref_added_sorted']
        if c['truncated']:
            for later in calls:
                later['diverged'] = True       # from here the reference's store holds its own arbitrary choice
    tokens, rounds = dev_set.read()
    got = [''.join(ALPHABET[t] for t in row) for row in tokens.cpu().tolist()]
    assert got == store.seqs and rounds.cpu().tolist() == store.rounds
    return len(got)


def test_acquisition_set_follows_update_data_scripts(rt):
    sc = json.load(open(os.path.join(GOLDEN, 'acquire_update.json')))
    for name, calls in sc.items():
        if name != 'landscape':
            _run_script(rt, calls, 12)


def test_acquisition_set_large_random_against_restatement(rt):
    rng = np.random.default_rng(1)
    L, N = 119, 16384
    dev_set = rt.AcquisitionSet(L, capacity=200000, max_batch=N)
    store = acq.SequenceStore()
    parent = rng.integers(0, 20, L)
    has_seqs, round_start, cur_round = False, None, 0
    for call, (BO, rnd, uniq, n_needed) in enumerate([(False, 0, True, 10 ** 6), (False, 0, True, 10 ** 6), (False, 1, True, 5000),
                                                     (True, 1, True, 10 ** 6), (True, 2, True, 777), (False, 3, False, 10 ** 6),
                                                     (False, 3, True, 10 ** 6)]):
        # few mutations of one parent: many duplicates inside the batch and against earlier batches
        x = np.tile(parent, (N, 1))
        pos = rng.integers(0, 6, (N, 2))
        x[np.arange(N)[:, None], pos] = rng.integers(0, 20, (N, 2))
        if rnd != cur_round:
            cur_round, round_start = rnd, None
        scope = 0 if not has_seqs else 1 if BO else 2 if (round_start is not None and uniq) else 0
        before = len(dev_set)
        kept, adm = dev_set.update(torch.from_numpy(x).cuda(), n_needed, scope, rnd)
        has_seqs = True
        round_start = before if round_start is None else round_start
        idx, adm_ref = store.update([r.tobytes() for r in x], n_needed, BO=BO, round=rnd, unique_only=uniq)
        assert adm == adm_ref and kept.cpu().tolist() == idx, call
    tokens, rounds = dev_set.read()
    assert [r.tobytes() for r in tokens.cpu().numpy()] == store.seqs
    assert rounds.cpu().tolist() == store.rounds
    # properties: under BO nothing is admitted twice; a replayed batch adds nothing
    kept, adm = dev_set.update(torch.from_numpy(x).cuda(), 10 ** 6, 1, 9)
    assert kept.numel() == 0 and adm == 0


def test_acquisition_set_survives_many_truncated_rounds(rt):
    """Rows cut off by n_needed give their slot up as a tombstone; tombstones are swept (the table is refilled from the
    store) before they can fill it.  60 updates of up to 64 rows with n_needed = 3 make about 3000 tombstones in a
    1024-slot table; every update must still agree with the restatement, including rows that come back later and
    the round-scoped filter (scope 2) after a sweep."""
    L = 20
    rng = np.random.default_rng(3)
    pool = [''.join(ALPHABET[t] for t in rng.integers(0, 20, L)) for _ in range(6000)]
    calls, seen = [], 0
    for it in range(60):
        fresh = pool[seen:seen + 56]
        seen += 56
        batch = fresh + [pool[i] for i in rng.integers(0, seen, 8)]     # some rows met before
        rng.shuffle(batch)
        calls.append(dict(batch=batch, n_needed=3, BO=(it % 5 != 4), round=it // 10, unique_only=True, truncated=True,
                          diverged=True))
    n = _run_script(rt, calls, L, capacity=400, max_batch=64)
    assert n == 180


def test_acquisition_set_hash_collisions_are_resolved_by_token_compare(rt):
    # a tiny table (capacity 30 -> 1024 slots is the floor), many sequences differing in one late position:
    # linear probing chains, tombstones from truncation, re-admission of a truncated sequence
    L = 300
    dev_set = rt.AcquisitionSet(L, capacity=600, max_batch=512)
    base = torch.zeros(512, L, dtype=torch.int64)
    base[:, -1] = torch.arange(512) % 31
    base[:, -2] = torch.arange(512) // 31
    kept, adm = dev_set.update(base.cuda(), 100, 1, 0)            # store empty but scope 1: 512 unique, 100 kept
    assert adm == 512 and kept.cpu().tolist() == list(range(100))
    kept, adm = dev_set.update(base.cuda(), 1000, 1, 0)           # the 412 cut off before are admissible again
    assert adm == 412 and kept.cpu().tolist() == list(range(100, 512))
    kept, adm = dev_set.update(base.flip(0).cuda(), 1000, 1, 1)
    assert adm == 0 and len(dev_set) == 512
    with pytest.raises(Exception):
        dev_set.update(torch.ones(512, L, dtype=torch.int64).cuda(), 1000, 0, 1)   # 512 more rows do not fit in 600


def test_dataset_dropin_update_data(rt):
    from dataset.protein import ProteinPredictorDataset
    from sgpo_amd import weights as W
    from sgpo_amd.tokenizer import Tokenizer
    from oracle import oracle_ensemble as oe
    sc = json.load(open(os.path.join(GOLDEN, 'acquire_update.json')))
    # 4-site landscape branch against the reference's y
    ls = sc['landscape']
    import pandas as pd
    with tempfile.TemporaryDirectory() as d:
        pd.DataFrame({'Combo': ls['combos'], 'fitness': ls['fitness']}).to_csv(os.path.join(d, 'fitness.csv'), index=False)
        dc = dict(name='GB1', data_path=os.path.join(d, 'fitness.csv'), full_seq=ls['seqs'][0], seq_len=8, alphabet_size=31,
                  residues=[r + 1 for r in ls['residues']])
        ds = ProteinPredictorDataset(dc, Tokenizer(), capacity=1024, max_batch=256)
        n = ds.update_data(ls['seqs'], 1000)
        assert n == len(ls['seqs']) and ds.seqs == ls['seqs']
        assert np.array_equal(ds.y.numpy(), np.array(ls['y'], dtype=np.float32))
    # ensemble branch: strings and tokens give the same store and the restatement's fitness
    L = 24
    rng = np.random.default_rng(3)
    parent = ''.join(ALPHABET[t] for t in rng.integers(0, 20, L))
    states = W.oracle_ensemble_synthetic_state(seed=5, n_members=4, n_res=L, hidden=32)
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, 'models'))
        for m, sd in enumerate(states):
            torch.save(sd, os.path.join(d, 'models', f'model{m}.pth'))
        os.makedirs(os.path.join(d, 'data', 'CreiLOV'))
        open(os.path.join(d, 'data', 'CreiLOV', 'parent.fasta'), 'w').write('>p\n' + parent + '\n')
        cwd = os.getcwd()
        os.chdir(d)
        try:
            dc = dict(name='CreiLOV', data_path=None, full_seq=parent, seq_len=L, alphabet_size=31, residues=None,
                      oracle_model_path=os.path.join(d, 'models'))
            ds_s = ProteinPredictorDataset(dc, Tokenizer(), capacity=1024, max_batch=256)
            ds_t = ProteinPredictorDataset(dc, Tokenizer(), capacity=1024, max_batch=256)
            store = acq.SequenceStore()
            for rnd, BO in ((0, False), (0, False), (1, True)):
                x = np.tile([ALPHABET.index(c) for c in parent], (64, 1))
                pos = rng.integers(0, 3, (64, 1))
                x[np.arange(64)[:, None], pos] = rng.integers(0, 20, (64, 1))
                seqs = [''.join(ALPHABET[t] for t in r) for r in x]
                n_s = ds_s.update_data(seqs, 40, BO=BO, round=rnd)
                n_t = ds_t.update_data(torch.from_numpy(x).cuda(), 40, BO=BO, round=rnd)
                idx, _ = store.update(seqs, 40, BO=BO, round=rnd)
                assert n_s == n_t == len(idx)
            assert ds_s.seqs == ds_t.seqs == store.seqs
            assert ds_s.round_seqs == store.round_seqs
            want = oe.ensemble_fitness(states, store.seqs, parent, 'CreiLOV').float()
            torch.testing.assert_close(ds_s.y, want, rtol=2e-5, atol=1e-7)
            assert torch.equal(ds_s.y, ds_t.y)
            df = ds_s.summary_df
            assert list(df['sequence']) == store.seqs and list(df['round']) == store.rounds
        finally:
            os.chdir(cwd)
