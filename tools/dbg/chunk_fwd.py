"""Experiment: one DiT forward over 4096 sequences as one call vs n sub-batches run back to back (intermediates of a
sub-batch fit the 256-MB last-level cache between producer and consumer kernels)."""
import sys, os, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from sgpo_amd import runtime as rt, weights as W
B = 4096
rt.init('cuda:0')
net = rt.DitDenoiser(W.dit_synthetic_state(seed=0))
g = torch.Generator().manual_seed(1)
x = torch.randint(0, 31, (B, 119), generator=g).cuda()
sigma = (torch.rand(B, generator=g) * 3).cuda()
ref = net.forward(x, sigma).clone()
for n in (1, 2, 4, 8, 16, 32):
    c = B // n
    outs = None
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = [net.forward(x[i * c:(i + 1) * c], sigma[i * c:(i + 1) * c]).clone() for i in range(n)]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    same = torch.equal(torch.cat(outs), ref)
    print(f'{n:3d} sub-batches of {c:5d}: {dt * 1e3:7.2f} ms  bit-identical {same}', flush=True)
