"""Per-kernel times of one ByteNet forward at the config-3 batch (B x 389 rows), HIP events on the launch stream.
usage: bytenet_bench.py [B] [reps]   (SGPO_HIP_LIB selects a diagnostic build for A/B runs)"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from sgpo_amd import runtime as rt, weights as W, _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rt.init('cuda:0')
lib = L.load()
net = rt.BytenetDenoiser(W.bytenet_synthetic_state(seed=0))
g = torch.Generator().manual_seed(1)
x = torch.randint(0, 26, (B, 389), generator=g).cuda()
net.forward(x, 250)
torch.cuda.synchronize()
L.check(lib.sgpo_prof_enable(1, 4096), 'prof')
for _ in range(reps):
    net.forward(x, 250)
torch.cuda.synchronize()
n = L.PROF_KINDS
ms = (C.c_double * n)(); cnt = (C.c_int64 * n)(); work = (C.c_double * n)()
L.check(lib.sgpo_prof_read(n, ms, cnt, work), 'read')
lib.sgpo_prof_enable(0, 0)
tot = sum(ms)
print(f'--- total {tot / reps:.2f} ms per forward')
for k in range(n):
    if cnt[k]:
        nm = lib.sgpo_prof_kind_name(k).decode()
        us = 1e3 * ms[k] / cnt[k]
        print(f'  {nm:36s} n={cnt[k]:4d} avg {us:9.1f} us  {work[k] / cnt[k] / (us * 1e-6) / 1e12:8.1f} T(FLOP|B)/s  share {ms[k] / tot:.3f}')
