"""Row f-3 timing: optimiser steps per second of the lockstep ensemble trainer (10 members, CreiLOV-sized MLPModel,
batch 64) against the same step written with torch autograd + torch.optim.Adam on the same GPU (one member at a
time, which is what one of the reference's spawned training processes does)."""
import sys, os, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import sgpo_amd
sgpo_amd.install_dropin()
from sgpo_amd import runtime as rt, weights as W
from problem.protein import MLPModel
rt.init('cuda:0')
E, B, Lq, steps = 10, 64, 119, 300
states = [W.mlp_synthetic_state(seed=e, n_residues=Lq) for e in range(E)]
tr = rt.MlpEnsembleTrainer(states, list(range(Lq)))
g = torch.Generator(device='cuda:0').manual_seed(0)
x = torch.randint(0, 31, (E, B, Lq), device='cuda:0', generator=g)
t = torch.randint(0, 128, (E, B), device='cuda:0', generator=g).float()
y = torch.randn(E, B, device='cuda:0', generator=g)
for _ in range(10):
    tr.step(x, t, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(x, t, y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f'fused trainer: {dt * 1e6:.1f} us per step for {E} members ({E / dt:.0f} member-steps/s)')
dc = dict(alphabet_size=31, residues=None, full_seq='A' * Lq)
m = MLPModel(dc, dict(hidden_dim=64)).cuda()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
crit = torch.nn.MSELoss()
def one():
    opt.zero_grad()
    loss = crit(m(x[0], t[0]).squeeze(), y[0])
    loss.backward()
    opt.step()
for _ in range(10):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    one()
torch.cuda.synchronize()
dt2 = (time.perf_counter() - t0) / steps
print(f'torch autograd + Adam on the same GPU: {dt2 * 1e6:.1f} us per step for 1 member ({1 / dt2:.0f} member-steps/s)')
print(f'ratio per member-step: {dt2 * E / dt:.1f}x')
