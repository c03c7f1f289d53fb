"""ByteNet (D3PM denoiser) forward at the config-3 per-GPU shard (2048 x 389 tokens): per-kernel times from the
library's own HIP-event profiler."""
import sys, os, ctypes as C, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from sgpo_amd import runtime as rt, weights as W, _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
Lq = int(sys.argv[2]) if len(sys.argv) > 2 else 389
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rt.init('cuda:0')
lib = L.load()
net = rt.BytenetDenoiser(W.bytenet_synthetic_state(seed=0))
g = torch.Generator().manual_seed(1)
x = torch.randint(0, 26, (B, Lq), generator=g).cuda()
t = torch.randint(1, 500, (B,), generator=g).cuda()
out = net.forward(x, t)
torch.cuda.synchronize()
L.check(lib.sgpo_prof_enable(1, reps * 200 + 64), 'prof')
t0 = time.perf_counter()
for _ in range(reps):
    out = net.forward(x, t, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
n = L.PROF_KINDS
ms = (C.c_double * n)(); cnt = (C.c_int64 * n)(); work = (C.c_double * n)()
L.check(lib.sgpo_prof_read(n, ms, cnt, work), 'read')
lib.sgpo_prof_enable(0, 0)
print(f'forward {dt * 1e3:.2f} ms  ({B * Lq / dt / 1e6:.2f} Mtoken/s, {75.6e6 * B * Lq / dt / 1e12:.0f} TFLOP/s at 75.6 MFLOP/token)')
tot = sum(ms)
for k in range(n):
    if cnt[k]:
        nm = lib.sgpo_prof_kind_name(k).decode()
        per = ms[k] / cnt[k]
        rate = work[k] / cnt[k] / (per * 1e-3)
        print(f'{nm:40s} {cnt[k] // reps:4d}/fwd  {per * 1e3:9.1f} us  {ms[k] / tot * 100:5.1f} %  {rate / 1e12:8.1f} T(flop|B)/s')
