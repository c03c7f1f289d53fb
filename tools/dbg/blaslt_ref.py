"""Diagnostic: time the vendor bf16 GEMM (torch.mm -> hipBLASLt) on the denoiser's four shapes (no fused epilogue)."""
import torch, time
M = 4096 * 119
for name, K, N in (("qkv", 512, 1536), ("out", 512, 512), ("fc1", 512, 2048), ("fc2", 2048, 512)):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        torch.mm(a, w.t(), out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        torch.mm(a, w.t(), out=c)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"{name}: {us:.0f} us  {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s", flush=True)
    del a, w, c
