"""One ByteNet forward at the config-3 batch (B x 389 rows) for PMC / kernel-trace runs."""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from sgpo_amd import runtime as rt, weights as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rt.init('cuda:0')
net = rt.BytenetDenoiser(W.bytenet_synthetic_state(seed=0))
g = torch.Generator().manual_seed(1)
x = torch.randint(0, 26, (B, 389), generator=g).cuda()
for _ in range(reps):
    out = net.forward(x, 250)
torch.cuda.synchronize()
print('ok', float(out.float().abs().mean()))
