"""CPU: host-side bodies of the drop-in that must consume numpy's / torch's global RNG exactly as the reference does
(sampling/base.py:40-63, sampling/daps.py:75-114, dataset/protein.py:92-135), checked against outputs of the
reference's own functions (tests/golden/acquire_project.npz; tools/refgen/make_goldens.py gen_acquire) and against the
oracle restatement."""
import os
import types

import numpy as np
import torch

import sgpo_amd

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
ALPHABET = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'


def _algo(L, full_s, n_max, residues=None):
    sgpo_amd.install_dropin()
    from sampling.base import Algo
    from sgpo_amd.tokenizer import Tokenizer

    class _A(Algo):
        def inference(self, observation=None, num_samples=1, **kw):
            raise NotImplementedError

    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=L, full_seq=full_s, residues=residues)
    return _A(net=types.SimpleNamespace(tokenizer=Tokenizer()), data_config=dc, n_max_mutations=n_max, device='cpu')


def test_project_sequences_consumes_numpy_rng_like_the_reference(monkeypatch):
    """np.random.choice is replaced by the reader of the noise arrays that the golden was made with: one call per special
    character met while walking each original string (all strings first), then one call per string for the
    max-mutation cut; the strings that come out are the reference's."""
    g = np.load(os.path.join(GOLD, 'acquire_project.npz'))
    for case in 'abcd':
        x, full, u_special, keys = g[f'{case}_x'], g[f'{case}_full'], g[f'{case}_u_special'], g[f'{case}_keys']
        n_max = int(g[f'{case}_n_max'])
        n_max = None if n_max < 0 else n_max
        N, L = x.shape
        algo = _algo(L, ''.join(ALPHABET[t] for t in full), n_max)
        q1 = [ALPHABET[min(int(np.float32(u_special[n, x[n, j] - 20]) * np.float32(20)), 19)]
              for n in range(N) for j in range(L) if x[n, j] >= 20]
        state = dict(i1=0, n2=0, order=[])

        def choice(a, size=None, replace=True, p=None):
            if size is None:
                assert list(a) == list(ALPHABET[:20])
                assert state['n2'] == 0                       # every step-1 draw precedes the first step-2 draw
                v = q1[state['i1']]
                state['i1'] += 1
                return v
            assert replace is False
            n = state['n2']
            state['n2'] += 1
            return np.array(sorted(list(a), key=lambda j: (float(keys[n, j]), j))[:size], dtype=np.int64)
        monkeypatch.setattr(np.random, 'choice', choice)
        res = algo.project_sequences([''.join(ALPHABET[t] for t in row) for row in x])
        monkeypatch.undo()
        assert state['i1'] == len(q1) and state['n2'] == (N if n_max is not None else 0), case
        y = np.array([[ALPHABET.index(c) for c in r] for r in res])
        assert np.array_equal(y, g[f'{case}_y']), case


def test_algo_attributes_match_the_reference_contract():
    a = _algo(12, 'MKGYFGPYGGQY', None, residues=[2, 5, 6, 11])
    assert a.residues == [1, 4, 5, 10] and a.n_residues == 4 and a.all_residues == list(range(12))   # base.py:30
    assert a.mask.dtype == torch.int32 and a.mask.tolist() == [0, 1, 0, 0, 1, 1, 0, 0, 0, 0, 1, 0]
    assert a.full_seq.dtype == torch.int32 and a.full_seq.tolist() == [ALPHABET.index(c) for c in 'MKGYFGPYGGQY']
    assert a.alphabet == list(ALPHABET[:20]) and a.special_tokens == list(ALPHABET[20:])
    b = _algo(5, 'ACDEF', None)
    assert b.residues == [0, 1, 2, 3, 4] and int(b.mask.sum()) == 5


def test_generic_metropolis_hastings_matches_the_oracle_chain():
    """The fallback MH loop for predictors without a fused kernel (e.g. a GP draw) against the restatement of
    sampling/daps.py:75-114, same torch seed -> same proposals, same acceptances, same tokens."""
    sgpo_amd.install_dropin()
    from oracle import daps as odaps
    from oracle import mlp as omlp
    from sampling.daps import DAPS
    from sgpo_amd import weights as W
    from sgpo_amd.tokenizer import Tokenizer
    L, S, N, steps = 30, 26, 5, 80
    res1 = [3, 4, 9, 17, 25]
    msd = W.mlp_synthetic_state(seed=6, n_residues=len(res1))
    pred = omlp.Predictor(msd, res1)
    net = types.SimpleNamespace(tokenizer=Tokenizer(), S=S, length=L, timestep=500, _dev=None)
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=L, full_seq='M' * L, residues=res1)
    algo = DAPS.__new__(DAPS)
    from sampling.base import Algo
    Algo.__init__(algo, net, pred, dc, device='cpu')
    algo.beta, algo.max_dist = 1., 1
    g = torch.Generator().manual_seed(4)
    x0hat = torch.randint(0, S, (N, L), generator=g)
    moved = []
    for sigma in (torch.tensor(0.05), torch.tensor(3.0), torch.tensor(float('inf'))):
        torch.manual_seed(21)
        r_idx = torch.empty(steps, N, dtype=torch.int64)
        r_val = torch.empty(steps, N, dtype=torch.int64)
        r_u = torch.empty(steps, N)
        for i in range(steps):
            r_idx[i], r_val[i], r_u[i] = torch.randint(len(res1), (N,)), torch.randint(S, (N,)), torch.rand(N)
        want = odaps.metropolis_hasting(x0hat, pred.log_likelihood, sigma, steps, 4., 1., algo.residues, S, L, r_idx, r_val, r_u)
        torch.manual_seed(21)
        got = algo._metropolis_hasting_generic(x0hat, pred, None, sigma, steps, 4.)
        assert torch.equal(got, want)
        moved.append(bool((got != x0hat).any()))
    assert any(moved)


def test_sample_random_variants_draw_order(monkeypatch):
    """dataset/protein.py:108-135: per variant one choice of the mutated sites (when n_max_mutations is set), then one
    letter per site, in site order."""
    sgpo_amd.install_dropin()
    from dataset.protein import ProteinPredictorDataset
    ds = ProteinPredictorDataset.__new__(ProteinPredictorDataset)
    ds.full_seq_string, ds.residues, ds.n_residues = 'MKGYFGPYGGQY', [1, 4, 5, 10], 4
    calls = []
    real = np.random.choice

    def spy(a, size=None, replace=True, p=None):
        calls.append((len(a), size, replace))
        return real(a, size=size, replace=replace, p=p)
    monkeypatch.setattr(np.random, 'choice', spy)
    ds.n_max_mutations = None
    np.random.seed(0)
    v = ds.sample_random_variants(3)
    assert calls == [(20, None, True)] * 12 and len(v) == 3
    assert all(len(s) == 12 and all(s[j] == 'MKGYFGPYGGQY'[j] for j in range(12) if j not in ds.residues) for s in v)
    np.random.seed(0)
    want = []
    for _ in range(3):                                # the same stream drawn by hand
        s = list('MKGYFGPYGGQY')
        for r in ds.residues:
            s[r] = real(list('ACDEFGHIKLMNPQRSTVWY'))
        want.append(''.join(s))
    assert v == want
    calls.clear()
    ds.n_max_mutations = 2
    ds.sample_random_variants(2)
    assert calls == [(4, 2, False), (20, None, True), (20, None, True)] * 2
    assert ds.get_full_sequence('WWWW') == 'MWGYWWPYGGWY'


def test_bench_workloads_count_their_own_units():
    """bench.py's `value` is units() * steps / time: every workload class states its own units per step (config 1 once
    inherited config 5's ten stacked draws and reported ten times its rate)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('bench_contract', os.path.join(os.path.dirname(os.path.dirname(__file__)), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    classes = [v for k, v in vars(bench).items() if isinstance(v, type) and k.startswith('_Config')]
    assert len(classes) >= 4
    for c in classes:
        assert 'units' in c.__dict__ and 'describe' in c.__dict__, c.__name__

    class _Fake:
        world, B, DRAWS, steps_per_run = 2, 64, 10, 50
    assert bench._Config1.units(_Fake) == 128
    assert bench._Config5.units(_Fake) == 1280
