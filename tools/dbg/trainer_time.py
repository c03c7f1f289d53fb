import sys, os, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from sgpo_amd import runtime as rt, weights as W
rt.init('cuda:0')
E_m, B_t = 10, 64
tr = rt.MlpEnsembleTrainer([W.mlp_synthetic_state(seed=50 + e, n_residues=119) for e in range(E_m)], list(range(119)))
g = torch.Generator(device='cuda').manual_seed(4)
x = torch.randint(0, 31, (50, E_m, B_t, 119), device='cuda', generator=g)
t = torch.randint(0, 128, (50, E_m, B_t), device='cuda', generator=g).float()
y = torch.randn(50, E_m, B_t, device='cuda', generator=g)
tr.steps(x, t, y); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): tr.steps(x, t, y)
torch.cuda.synchronize()
print('ms per step', (time.perf_counter() - t0) * 1e3 / 500)
