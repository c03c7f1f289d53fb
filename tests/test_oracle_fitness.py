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
