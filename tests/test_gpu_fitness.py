"""GPU parity of the fitness-oracle ensemble through the C ABI and the
drop-in `fitness.inference_oracle.inference_oracle`: goldens from the reference's own
oracle/inference_oracle.py, the CPU restatement on larger random batches, edge cases."""
import os
import tempfile

import pytest
import torch

from conftest import load_golden
from test_oracle_fitness import _case

pytestmark = pytest.mark.gpu
AA20 = 'ACDEFGHIKLMNPQRSTVWY'


@pytest.fixture(scope='module')
def rt():
    import sgpo_amd
    sgpo_amd.install_dropin()
    from sgpo_amd import runtime
    runtime.init('cuda:0')
    return runtime


def test_dropin_inference_oracle_against_reference_goldens(rt):
    from fitness.inference_oracle import inference_oracle
    g = load_golden('oracle_ens')
    for protein in ('CreiLOV', 'TrpB'):
        seqs, parent, states = _case(g, protein)
        with tempfile.TemporaryDirectory() as d:
            for m, sd in enumerate(states):
                torch.save(sd, os.path.join(d, f'model{m}.pth'))
            loss, y = inference_oracle(seqs, protein=protein, model_path=d, full_seq=parent)
            _, y0 = inference_oracle(seqs, protein=protein, model_path=d, impose_penalty=False, full_seq=parent)
        assert y.dtype == torch.float64 and y.shape == (len(seqs),)
        # fp32 gather-sum on the device vs fp32 BLAS in the reference: summation order only
        torch.testing.assert_close(y, g[f'{protein}_y'], rtol=2e-5, atol=1e-7)
        torch.testing.assert_close(y0, g[f'{protein}_y_nopenalty'], rtol=2e-5, atol=1e-7)
        assert abs(float(loss) - float((g[f'{protein}_y'] ** 2).mean())) < 1e-6


def test_ensemble_kernel_vs_restatement_large_batch(rt):
    from oracle import oracle_ensemble as oe
    from sgpo_amd import weights as W
    g = torch.Generator().manual_seed(9)
    for n_res, members, protein, N in ((119, 10, 'CreiLOV', 700), (15, 20, 'TrpB', 1500), (7, 3, 'CreiLOV', 5)):
        states = W.oracle_ensemble_synthetic_state(seed=4, n_members=members, n_res=n_res)
        tok = torch.randint(0, 20, (N, n_res), generator=g)
        parent = torch.randint(0, 20, (n_res,), generator=g)
        tok[: N // 2] = torch.where(torch.rand(N // 2, n_res, generator=g) < 0.3, tok[: N // 2], parent)  # near the parent
        seqs = [''.join(AA20[int(t)] for t in row) for row in tok]
        par_s = ''.join(AA20[int(t)] for t in parent)
        ens = rt.FitnessEnsemble(states)
        cutoff, rate = (3, 0.9) if n_res < 20 else oe.PENALTY[protein]
        oe_pen = dict(oe.PENALTY)
        oe.PENALTY[protein] = (cutoff, rate)
        try:
            ref = oe.ensemble_fitness(states, seqs, par_s, protein)
        finally:
            oe.PENALTY.update(oe_pen)
        got = ens.forward(tok, parent, cutoff, rate).cpu()
        torch.testing.assert_close(got, ref, rtol=2e-5, atol=1e-7)
        ref0 = oe.ensemble_fitness(states, seqs, par_s, protein, impose_penalty=False)
        torch.testing.assert_close(ens.forward(tok, None).cpu(), ref0, rtol=2e-5, atol=1e-7)
        assert float(ref.min()) >= 0.0 and float(got.min()) >= 0.0


def test_letter_outside_alphabet_raises_like_the_reference(rt):
    from fitness.inference_oracle import inference_oracle
    from sgpo_amd import weights as W
    states = W.oracle_ensemble_synthetic_state(seed=1, n_members=2, n_res=6)
    ens = rt.FitnessEnsemble(states)
    tok = torch.tensor([[0, 1, 2, 3, 4, 25]])
    with pytest.raises(ValueError):
        ens.forward(tok, None)
    assert ens.forward(torch.tensor([[0, 1, 2, 3, 4, 5]]), None).shape == (1,)     # the flag was cleared
    with tempfile.TemporaryDirectory() as d:
        for m, sd in enumerate(states):
            torch.save(sd, os.path.join(d, f'model{m}.pth'))
        with pytest.raises(ValueError):
            inference_oracle(['ACDEFB'], protein='CreiLOV', model_path=d, full_seq='ACDEFG')   # 'B' is not one of the 20
    with pytest.raises(Exception):
        ens.forward(torch.zeros(2, 5, dtype=torch.int64), None)                     # wrong number of residues
