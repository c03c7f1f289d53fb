"""CPU: the fitness-ensemble restatement (oracle/oracle_ensemble.py) against goldens captured from the
reference's own oracle/inference_oracle.py (tools/refgen/make_goldens.py::gen_oracle_ens), which at
generation time also checked the restatement against the reference on the checkpoints it ships."""
import torch

from conftest import load_golden


def _case(g, protein):
    from sgpo_amd import weights as W
    seqs = [str(s) for s in g[f'{protein}_seqs']]
    parent = str(g[f'{protein}_parent'])
    states = W.oracle_ensemble_synthetic_state(seed=int(g[f'{protein}_seed']), n_members=int(g[f'{protein}_members']),
                                               n_res=len(seqs[0]))
    return seqs, parent, states


def test_restatement_matches_reference_goldens():
    from oracle import oracle_ensemble as oe
    g = load_golden('oracle_ens')
    for protein in ('CreiLOV', 'TrpB'):
        seqs, parent, states = _case(g, protein)
        y = oe.ensemble_fitness(states, seqs, parent, protein)
        torch.testing.assert_close(y, g[f'{protein}_y'], rtol=1e-6, atol=1e-8)
        y0 = oe.ensemble_fitness(states, seqs, parent, protein, impose_penalty=False)
        torch.testing.assert_close(y0, g[f'{protein}_y_nopenalty'], rtol=1e-6, atol=1e-8)
    # the penalty is active in the CreiLOV case (some sequences are > 70 mutations from the parent)
    assert float((g['CreiLOV_y'] - g['CreiLOV_y_nopenalty']).abs().max()) > 0
    # residue combos are compared with the first len(combo) letters of the parent (zip truncation)
    assert oe.penalty(71, 70, 0.99) == 0.99 and oe.penalty(70, 70, 0.99) == 1


def test_restatement_on_the_shipped_ensemble_scores_the_reference_free_run():
    """The reference's 64-sequence guided free run (free_run_real.npz) was scored by the reference's inference_oracle
    on its shipped CreiLOV ensemble; the restatement on the exported copy of those weights (oracle/_ref/CreiLOV_ens,
    written by __graft_entry__.build() where the reference checkout exists) reproduces those numbers."""
    import os

    import pytest

    from conftest import ROOT
    from oracle import oracle_ensemble as oe
    ens = os.path.join(ROOT, 'oracle', '_ref', 'CreiLOV_ens')
    if not os.path.isdir(ens):
        pytest.skip('oracle/_ref/CreiLOV_ens missing (no reference checkout at build time)')
    g = load_golden('free_run_real')
    alphabet = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'
    seqs = [''.join(alphabet[int(t)] for t in row) for row in g['repaired']]
    parent = bytes(g['full_seq'].numpy().tolist()).decode()
    states = [torch.load(os.path.join(ens, str(f)), map_location='cpu', weights_only=True) for f in g['member_files']]
    y = oe.ensemble_fitness(states, seqs, parent, 'CreiLOV')
    torch.testing.assert_close(y.double(), torch.as_tensor(g['fitness']).double(), rtol=1e-6, atol=1e-8)
    y0 = oe.ensemble_fitness(states, seqs, parent, 'CreiLOV', impose_penalty=False)
    torch.testing.assert_close(y0.double(), torch.as_tensor(g['fitness_nopenalty']).double(), rtol=1e-6, atol=1e-8)
    assert float(y.mean()) > 0 and float((y0 - y).abs().max()) > 0      # the Hamming penalty is active on these samples
