"""Row f-2 on the CPU: the restatement (oracle/acquisition.py) against the outputs of the reference's own
project_sequences / update_data (tests/golden/acquire_project.npz, acquire_update.json; generator:
tools/refgen/make_goldens.py gen_acquire)."""
import json
import os

import numpy as np

from oracle import acquisition as acq

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_project_tokens_matches_reference_outputs():
    g = np.load(os.path.join(GOLD, 'acquire_project.npz'))
    for case in 'abcd':
        n_max = int(g[f'{case}_n_max'])
        y = acq.project_tokens(g[f'{case}_x'], g[f'{case}_full'], 20, g[f'{case}_u_special'],
                               None if n_max < 0 else n_max, g[f'{case}_keys'])
        assert np.array_equal(y, g[f'{case}_y']), case
        assert (y < 20).all()
        if n_max >= 0:
            assert ((y != g[f'{case}_full']).sum(axis=1) <= n_max).all()


def test_update_filter_matches_reference_as_sets():
    sc = json.load(open(os.path.join(GOLD, 'acquire_update.json')))
    for name, calls in sc.items():
        if name == 'landscape':
            continue
        store = acq.SequenceStore()
        for c in calls:
            idx, adm = store.update(c['batch'], c['n_needed'], BO=c['BO'], round=c['round'], unique_only=c['unique_only'])
            assert len(idx) == c['ref_returned'], name
            assert (adm > c['n_needed']) == c['truncated']
            added = [c['batch'][i] for i in idx]
            if c['truncated']:
                # both keep n_needed admissible sequences; which ones depends on set order in the reference
                assert len(set(added)) == len(added) or not c['unique_only']
                store.seqs = store.seqs[:len(store.seqs) - len(added)] + c['ref_added_in_order']
                store.round_seqs = store.round_seqs[:len(store.round_seqs) - len(added)] + c['ref_added_in_order']
            else:
                assert sorted(added) == c['ref_added_sorted'], name


def test_landscape_lookup_matches_reference():
    sc = json.load(open(os.path.join(GOLD, 'acquire_update.json')))['landscape']
    table = dict(zip(sc['combos'], sc['fitness']))
    combos = [''.join(s[r] for r in sc['residues']) for s in sc['seqs']]
    y = acq.landscape_lookup(table, combos)
    assert np.allclose(y, np.array(sc['y'], dtype=np.float32), rtol=0, atol=0)
    assert (y == 0).sum() >= 1      # the script holds combos the data frame lacks
