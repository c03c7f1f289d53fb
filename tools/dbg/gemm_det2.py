import sys, os, torch
sys.path.insert(0, '.')
from sgpo_amd import runtime as rt, weights as W
rt.init('cuda:0')
sd = W.dit_synthetic_state(seed=21)
net = rt.DitDenoiser(sd)
g = torch.Generator().manual_seed(9)
B, Lq = 36, 119
x = torch.randint(0, 31, (B, Lq), generator=g).cuda()
sigma = (torch.rand(B, generator=g) * 3).cuda()
a = net.forward(x, sigma).clone()
for i in (0, 16, 34):
    s = net.forward(x[i:i+2], sigma[i:i+2])
    print(i, (s - a[i:i+2]).abs().amax().item())
# raw GEMM (bf16 out path not exposed; fp32-out path): big vs small rows
A = torch.randn(4284, 512, generator=g).cuda(); Wt = (torch.randn(1536, 512, generator=g) / 22).cuda()
big = rt.gemm_bf16(A, Wt); sm = rt.gemm_bf16(A[:238], Wt)
print('gemm rows equal:', torch.equal(big[:238], sm), (big[:238] - sm).abs().max().item())
