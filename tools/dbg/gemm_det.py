import sys, os, torch
sys.path.insert(0, '.')
from sgpo_amd import runtime as rt, weights as W
rt.init('cuda:0')
sd = W.dit_synthetic_state(seed=21)
net = rt.DitDenoiser(sd)
g = torch.Generator().manual_seed(9)
B, Lq = 12, 119
x = torch.randint(0, 31, (B, Lq), generator=g).cuda()
sigma = (torch.rand(B, generator=g) * 3).cuda()
a = net.forward(x, sigma).clone()
b = net.forward(x, sigma).clone()
print('big twice equal:', torch.equal(a, b))
for i in range(0, B, 2):
    s = net.forward(x[i:i+2], sigma[i:i+2])
    d = (s - a[i:i+2]).abs().amax(dim=(1, 2))
    print(i, d.tolist())
s1 = net.forward(x[:2], sigma[:2]).clone(); s2 = net.forward(x[:2], sigma[:2]).clone()
print('small twice equal:', torch.equal(s1, s2))
# raw gemm: persistent vs rows computed in a small call
A = torch.randn(1428, 512, generator=g).cuda(); Wt = (torch.randn(1536, 512, generator=g) / 22).cuda()
big = rt.gemm_bf16(A, Wt); sm = rt.gemm_bf16(A[:238], Wt)
print('gemm rows equal:', torch.equal(big[:238], sm), (big[:238] - sm).abs().max().item())
