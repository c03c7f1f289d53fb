"""Secondary measurement: BASELINE config 4 (MDLM + NOS_discrete on TrpB: 200 sequences per guidance strength,
5 NOS steps of size 0.5 per reverse step, 127 reverse steps) on ONE GPU.  One JSON line like bench.py's.

    python tools/bench_nos.py [--batch 200] [--runs 3]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
ALPHABET = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'
TRPB_RES = [117, 118, 119, 162, 166, 182, 183, 184, 185, 186, 227, 228, 230, 231, 301]
L_TRPB, T_STEPS = 389, 128


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=200)
    ap.add_argument('--runs', type=int, default=3)
    a = ap.parse_args()
    import sgpo_amd
    sgpo_amd.install_dropin()
    from models.mdlm import MDLM
    from problem.protein_nos import NOSHead
    from sampling.NOS import NOS
    from sgpo_amd import _lib as L
    from sgpo_amd import weights as W
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(0)
    full = ''.join(ALPHABET[int(t)] for t in torch.randint(0, 20, (L_TRPB,), generator=g))
    cfg = dict(diffusion='absorbing_state', parameterization='subs', time_conditioning=False,
               model=dict(hidden_size=512, cond_dim=128, n_blocks=8, n_heads=8, length=1024))
    net = MDLM('synthetic', cfg, None, num_steps=T_STEPS, seq_len=L_TRPB, device=dev, state_dict=W.dit_synthetic_state(seed=0))
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=L_TRPB, full_seq=full, residues=TRPB_RES)
    torch.manual_seed(1)
    head = NOSHead(512).to(dev).eval()
    algo = NOS(net, dc, forward_op=head, num_nos_steps=5, nos_step_size=0.5, nos_stability_coef=0.1, device=dev)
    lib = L.load()
    net.reseed(3)
    algo.inference(num_samples=a.batch, verbose=False)
    torch.cuda.synchronize()
    L.check(lib.sgpo_prof_enable(1, a.runs * 127 * 80 + 64), 'prof_enable')
    t0 = time.perf_counter()
    for r in range(a.runs):
        net.reseed(4 + r)
        x = algo.inference(num_samples=a.batch, verbose=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.runs
    n = L.PROF_KINDS
    ms = (C.c_double * n)(); cnt = (C.c_int64 * n)(); work = (C.c_double * n)()
    L.check(lib.sgpo_prof_read(n, ms, cnt, work), 'prof_read')
    lib.sgpo_prof_enable(0, 0)
    tot = sum(ms)
    kernels = {lib.sgpo_prof_kind_name(k).decode(): dict(total_ms=round(ms[k], 2), launches=int(cnt[k]), avg_us=round(1e3 * ms[k] / cnt[k], 1),
                                                       share=round(ms[k] / tot, 4)) for k in range(n) if cnt[k]}
    print(json.dumps({'metric': 'guided sequences/sec (one guidance strength of config 4)', 'value': round(a.batch / dt, 2),
                      'unit': 'sequences/s', 'n_gpus': 1, 'ms_per_run': round(dt * 1e3, 1), 'dtype': 'bf16', 'data': 'synthetic',
                      'config': {'workload': 'MDLM + NOS_discrete on TrpB, L=389, %d sequences, 127 reverse steps x (1 denoiser forward + '
                                             '5 hidden-state optimisation steps)' % a.batch},
                      'gpu_busy_frac': round(tot * 1e-3 / (dt * a.runs), 4), 'kernels': kernels, 'checksum': int(x.sum())}))


if __name__ == '__main__':
    main()
