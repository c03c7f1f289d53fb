"""Time the four DiT GEMM shapes (M = B*119) through the C ABI; optional ablation via SGPO_GEMM_DBG.
usage: gemm_bench.py [B] [reps]   (SGPO_HIP_LIB=.../libsgpo_hip_ablate.so SGPO_GEMM_DBG=bits for ablation)"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from sgpo_amd import runtime as rt, weights as W, _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rt.init('cuda:0')
lib = L.load()
net = rt.DitDenoiser(W.dit_synthetic_state(seed=0))
g = torch.Generator().manual_seed(1)
x = torch.randint(0, 31, (B, 119), generator=g).cuda()
sigma = (torch.rand(B, generator=g) * 3).cuda()
for dbg in os.environ.get('DBG_LIST', '0').split(','):
    os.environ['SGPO_GEMM_DBG'] = dbg.split(':')[0]
    os.environ['SGPO_GEMM_STAGGER_US'] = dbg.split(':')[1] if ':' in dbg else '0'   # 'bits:us' = ablation bits : start stagger
    net.forward(x, sigma)
    torch.cuda.synchronize()
    L.check(lib.sgpo_prof_enable(1, 4096), 'prof')
    for _ in range(reps):
        net.forward(x, sigma)
    torch.cuda.synchronize()
    n = L.PROF_KINDS
    ms = (C.c_double * n)(); cnt = (C.c_int64 * n)(); work = (C.c_double * n)()
    L.check(lib.sgpo_prof_read(n, ms, cnt, work), 'read')
    lib.sgpo_prof_enable(0, 0)
    tot = sum(ms)
    print(f'--- dbg={dbg}  total {tot / reps:.2f} ms per forward')
    for k in range(n):
        if cnt[k]:
            nm = lib.sgpo_prof_kind_name(k).decode()
            us = 1e3 * ms[k] / cnt[k]
            tf = work[k] / cnt[k] / (us * 1e-6) / 1e12
            print(f'  {nm:24s} n={cnt[k]:4d} avg {us:9.1f} us  {tf:8.1f} T(FLOP|B)/s  share {ms[k] / tot:.3f}')
