"""gemm_q.hip (two 256-thread workgroups per CU, 256 x 128 tiles) against the persistent 256 x 256 kernel on the DiT shapes.
usage: gemm_q_bench.py [B] [reps]; SGPO_GEMMQ_MODE=1 runs its K loop alone."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from sgpo_amd import runtime as rt, _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rt.init('cuda:0')
lib = L.load()
M = B * 119
g = torch.Generator().manual_seed(1)


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


# correctness first, on a ragged shape: the bias-only epilogue against bf16(acc + bias) of the product GEMM (bit for bit while both
# used the 32x32x16 MFMA shape; the product kernels moved to 16x16x32 in round 3, so a few last-bit differences are expected now)
for (m, n, k) in ((1000, 256, 512), (777, 1536, 512), (4097, 2048, 512)):
    a = (torch.randn(m, k, generator=g) * 0.5).to(torch.bfloat16).cuda()
    w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(n, generator=g).cuda()
    out = torch.empty(m, n, dtype=torch.bfloat16, device='cuda')
    L.check(lib.sgpo_gemm_q_bf16(L.ptr(a), L.ptr(w), L.ptr(bias), L.ptr(out), m, n, k, 0, L.stream_ptr()), 'gemm_q')
    ref = (rt.gemm_bf16(a, w) + bias).to(torch.bfloat16)
    bad = int((out.view(torch.int16) != ref.view(torch.int16)).sum())
    og = torch.empty_like(out)
    L.check(lib.sgpo_gemm_q_bf16(L.ptr(a), L.ptr(w), L.ptr(bias), L.ptr(og), m, n, k, 3, L.stream_ptr()), 'gemm_q')
    rg = torch.nn.functional.gelu(rt.gemm_bf16(a, w) + bias, approximate='tanh')
    print(f'check {m}x{n}x{k}: bias epilogue differs in {bad} of {m * n} elements; gelu max |d| {float((og.float() - rg).abs().max()):.3e}')

for name, n, k, epi in (('qkv-like (bias)', 1536, 512, 0), ('fc1 + GELU', 2048, 512, 3), ('N=512 K=512 (bias)', 512, 512, 0), ('N=512 K=2048 (bias)', 512, 2048, 0)):
    a = (torch.randn(M, k, generator=g) * 0.5).to(torch.bfloat16).cuda()
    w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(n, generator=g).cuda()
    out = torch.empty(M, n, dtype=torch.bfloat16, device='cuda')
    us = timed(lambda: L.check(lib.sgpo_gemm_q_bf16(L.ptr(a), L.ptr(w), L.ptr(bias), L.ptr(out), M, n, k, epi, L.stream_ptr()), 'gemm_q'))
    print(f'gemm_q {name}: M={M} N={n} K={k}: {us:.1f} us, {2.0 * M * n * k / us / 1e6:.1f} TFLOP/s')
