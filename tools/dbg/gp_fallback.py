"""Cost of the generic-predictor path at config-5 batch sizes (f-4: GP Thompson draws).  botorch / gpytorch are not in
this image, so the predictor is a stand-in with the arithmetic of a `get_gp_samples` draw behind the reference's
`SampledDeterministicModel` (problem/protein.py:169-193, training/train_GP_classifier.py:103-111): the reference's own
`Embedder` form (one-hot -> Linear(31, 32) + time features, flattened: 119 x 32 = 3808 inputs), then 1000 random Fourier
features and a linear read-out, in fp64 as the GP is.  Times `Classifier_Guidance_Inpaint.inference(num_samples=B)` through
`_generic_jump_log_ratios` (materialised single-site jumps, models/mdlm.py:111-162) against the fused MLP path.
usage: gp_fallback.py [B]"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench  # noqa: E402


class RffDraw(torch.nn.Module):
    def __init__(self, n_res=119, alphabet=31, n_feat=1000, device='cuda:0'):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        h = alphabet + 1
        self.alphabet, self.h = alphabet, h
        self.emb = torch.nn.Linear(alphabet, h).to(device)
        self.W_t = (torch.randn(h // 2, generator=g) * 30).to(device)
        self.omega = (torch.randn(n_res * h, n_feat, generator=g, dtype=torch.float64) / math.sqrt(n_res * h)).to(device)
        self.phase = (torch.rand(n_feat, generator=g, dtype=torch.float64) * 2 * math.pi).to(device)
        self.w = torch.randn(n_feat, 1, generator=g, dtype=torch.float64).to(device)
        self.scale = math.sqrt(2.0 / n_feat)

    def forward(self, seq, t):
        x = self.emb(torch.nn.functional.one_hot(seq.long(), self.alphabet).float())
        proj = t[:, None] * self.W_t[None, :] * 2 * math.pi
        x = x + torch.cat([torch.sin(proj), torch.cos(proj)], dim=-1)[:, None, :]
        x = x.flatten(start_dim=-2).double()
        return (self.scale * torch.cos(x @ self.omega + self.phase) @ self.w).float()

    def __call__(self, inputs, t):
        return self.forward(inputs, t)

    def log_likelihood(self, inputs, y=None):
        return self(inputs, torch.zeros(inputs.shape[0], device=inputs.device)).squeeze(-1)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = 'cuda:0'
    net, pred, algo, sd, msd = bench.build(dev, B)
    draw = RffDraw(device=dev).eval()
    res = {}
    for name, op in (('fused MLP value function', pred), ('generic path, RFF GP draw (fp64, 1000 features)', draw)):
        algo.update_model(op)
        net.reseed(5)
        algo.inference(num_samples=B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(2):
            net.reseed(6 + r)
            algo.inference(num_samples=B)
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 2
        print(f'{name}: {res[name] * 1e3:.1f} ms per inference of {B} sequences ({res[name] / 127 * 1e3:.2f} ms per reverse step)')


if __name__ == '__main__':
    main()
