"""Secondary measurement: BASELINE config 3 (D3PM + daps_discrete on TrpB; 16384 sequences over 8 GPUs = 2048 per
GPU) on ONE GPU.  bench.py keeps the headline contract (config 2); this prints one JSON line of the same shape for
the DAPS workload.  A full run is 50 annealing steps x (19 ByteNet forwards + 1000 Metropolis-Hastings proposals);
`--anneal` of them are timed (every annealing step does identical work) and `value` is sequences per second of the
full 50-step run at that measured per-step time, with the measured fraction stated in `config`.

    python tools/bench_daps.py [--batch 2048] [--anneal 2]
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
L_TRPB, NUM_STEPS, ODE_STEPS, MH_STEPS = 389, 50, 20, 1000


def cpu_baseline(sd, msd, n_seq=16, n_fwd=6):
    """CPU restatement (oracle/, the reference's formulation) on a bounded sample (10-20 s of CPU work): n_fwd ByteNet
    forwards + posteriors and one 1000-proposal Metropolis-Hastings chain for n_seq sequences, scaled to the
    50 x (19 + 1) calls of a run."""
    from oracle import bytenet as obn
    from oracle import d3pm as od3pm
    from oracle import daps as odaps
    from oracle import mlp as omlp
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    S = 26
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, S, (n_seq, L_TRPB), generator=g)
    Q_bar, Q = od3pm.q_random_schedule(S, 500)
    sigma = od3pm.sigma_from_qbar(Q_bar, S)
    t0 = time.perf_counter()
    for t, tn in [(400 - 20 * k, 380 - 20 * k) for k in range(n_fwd)]:
        e = torch.empty(n_seq * L_TRPB, S, dtype=torch.float64).exponential_(1, generator=g)
        logits = obn.bytenet_forward(sd, x, torch.tensor([t] * n_seq))
        x = od3pm.p_sample(logits, x, t, tn, Q, Q_bar, sigma, S, e=e)
    t_fwd = (time.perf_counter() - t0) / n_fwd
    pred = omlp.Predictor(msd, TRPB_RES)
    r_idx = torch.randint(0, len(TRPB_RES), (MH_STEPS, n_seq), generator=g)
    r_val = torch.randint(0, S, (MH_STEPS, n_seq), generator=g)
    r_u = torch.rand(MH_STEPS, n_seq, generator=g)
    t0 = time.perf_counter()
    odaps.metropolis_hasting(x, pred.log_likelihood, sigma[300], MH_STEPS, 128.0, 1000, [r - 1 for r in TRPB_RES], S, L_TRPB,
                             r_idx, r_val, r_u)
    t_mh = time.perf_counter() - t0
    per_run = NUM_STEPS * ((ODE_STEPS - 1) * t_fwd + t_mh)
    return dict(value=n_seq / per_run, unit='sequences/s', cores=cores, kind='port',
                sample='%d sequences: %d denoiser forwards + posteriors (%.2f s each) and one 1000-proposal MH chain (%.2f s), '
                       'torch-CPU fp32 oracle, scaled to 50 x (19 forwards + 1 chain)' % (n_seq, n_fwd, t_fwd, t_mh))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--batch', type=int, default=2048)
    ap.add_argument('--anneal', type=int, default=2, help='annealing steps to time (of 50)')
    a = ap.parse_args()
    import sgpo_amd
    sgpo_amd.install_dropin()
    from models.d3pm import D3PM
    from problem.protein import ProteinPredictor
    from sampling.daps import DAPS
    from sgpo_amd import _lib as L
    from sgpo_amd import weights as W
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(0)
    full = ''.join(ALPHABET[int(t)] for t in torch.randint(0, 20, (L_TRPB,), generator=g))
    bn_sd = W.bytenet_synthetic_state(seed=0)
    net = D3PM('synthetic', L_TRPB, device=dev, state_dict=bn_sd)
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=L_TRPB, full_seq=full, residues=TRPB_RES)
    pred = ProteinPredictor(dc, dict(hidden_dim=64), device=dev)
    msd = W.mlp_synthetic_state(seed=1, n_residues=len(TRPB_RES))
    pred.model.load_state_dict(msd)
    pred.eval()
    lib = L.load()

    def run(n_anneal):
        algo = DAPS(net, forward_op=pred, num_steps=NUM_STEPS, ode_steps=ODE_STEPS, mh_steps=MH_STEPS, alpha=128., beta=1000,
                    data_config=dc, device=dev)
        algo.num_steps = n_anneal            # the first n_anneal steps of the 50-step schedule (time_steps is kept)
        net.reseed(7)
        return algo.inference(num_samples=a.batch, verbose=False)

    run(1)
    torch.cuda.synchronize()
    L.check(lib.sgpo_prof_enable(1, a.anneal * 3000 + 64), 'prof_enable')
    t0 = time.perf_counter()
    x = run(a.anneal)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = L.PROF_KINDS
    ms = (C.c_double * n)(); cnt = (C.c_int64 * n)(); work = (C.c_double * n)()
    L.check(lib.sgpo_prof_read(n, ms, cnt, work), 'prof_read')
    lib.sgpo_prof_enable(0, 0)
    tot = sum(ms)
    kernels = {}
    for k in range(n):
        if cnt[k]:
            kernels[lib.sgpo_prof_kind_name(k).decode()] = dict(total_ms=round(ms[k], 2), launches=int(cnt[k]),
                                                                avg_us=round(1e3 * ms[k] / cnt[k], 1), share=round(ms[k] / tot, 4),
                                                                work_per_launch=work[k] / cnt[k])
    per_step = dt / a.anneal
    gf = sum(work[k] for k in range(n) if lib.sgpo_prof_kind_name(k).decode().startswith('gemm'))
    gt = sum(ms[k] for k in range(n) if lib.sgpo_prof_kind_name(k).decode().startswith('gemm'))
    print(json.dumps({
        'metric': 'guided sequences/sec (one GPU shard of config 3)', 'value': round(a.batch / (per_step * NUM_STEPS), 3),
        'unit': 'sequences/s', 'n_gpus': 1, 's_per_annealing_step': round(per_step, 3), 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': 'D3PM (ByteNet 16 x d1024/h512, k5) + daps_discrete on TrpB, L=389, 15 residues, '
                               '%d sequences per GPU; %d of 50 annealing steps timed (each: 19 denoiser forwards + posterior, '
                               '1000 MH proposals, q_sample), value scaled to 50' % (a.batch, a.anneal)},
        'denoiser_gemm_tflops': round(gf / (gt * 1e-3) / 1e12, 1) if gt else None,
        'gpu_busy_frac': round(tot * 1e-3 / dt, 4), 'cpu_baseline': None if a.no_cpu_baseline else cpu_baseline(bn_sd, msd),
        'kernels': kernels, 'checksum': int(x.sum())}))


if __name__ == '__main__':
    main()
