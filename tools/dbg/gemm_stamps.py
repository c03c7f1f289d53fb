"""Per-section cycle totals of the persistent GEMM (diagnostic build: make variant NAME=stamp VFLAGS=-DSGPO_GEMM_STAMP).
usage: SGPO_HIP_LIB=.../libsgpo_hip_stamp.so SGPO_GEMM_STAMP_EPI=<epi> python gemm_stamps.py [B]"""
import sys, os, ctypes as C, torch, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from sgpo_amd import runtime as rt, weights as W, _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rt.init('cuda:0')
lib = L.load()
net = rt.DitDenoiser(W.dit_synthetic_state(seed=0))
g = torch.Generator().manual_seed(1)
x = torch.randint(0, 31, (B, 119), generator=g).cuda()
sigma = (torch.rand(B, generator=g) * 3).cuda()
for _ in range(3):
    net.forward(x, sigma)
torch.cuda.synchronize()
n = 256 * 8 * 8
buf = (C.c_ulonglong * n)()
raw = C.CDLL(os.environ['SGPO_HIP_LIB'])
raw.sgpo_gemm_stamps_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert raw.sgpo_gemm_stamps_read(buf, n) == 0
a = np.array(buf[:], dtype=np.float64).reshape(256, 8, 8)
names = ['dma_issue', 'reads+vmwait', 'lgkm_wait', 'barrier1', 'mfma_issue', 'epilogue', 'barrier2', 'other']
tot = a.sum(axis=2)
print('epi', os.environ.get('SGPO_GEMM_STAMP_EPI'), 'mean total cycles per wave %.0f' % tot.mean())
for grp, sl in (('waves0-3', slice(0, 4)), ('waves4-7', slice(4, 8))):
    m = a[:, sl, :].mean(axis=(0, 1))
    print(grp, ' '.join(f'{nm}={v:.0f}({v / m.sum():.2f})' for nm, v in zip(names, m)))
