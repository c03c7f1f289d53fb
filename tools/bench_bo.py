"""BASELINE config 5 on one GPU: one round of iterativeBO.py (iterativeBO.py:140-237) through the drop-ins --
train the value-function ensemble on the data gathered so far, then Thompson-sample: draw a random member, run guided
sampling for a small batch, score it with the fitness ensemble, de-duplicate against the store, until the round has its
new sequences.  MDLM (real DiT dims) on CreiLOV, synthetic weights and a synthetic fitness ensemble (no checkpoints
offline).  Prints one JSON line.

    python tools/bench_bo.py [--members 10] [--epochs 1000] [--init 500] [--per-round 100] [--draw 10]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ALPHABET = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'
CREILOV = ('MAGLRHTFVVADATLPDCPLVYASEGFYAMTGYGPDEVLGHNARFLQGEGTDPKEVQKIRDAIKKGEACSVRLLNYRKDGTPFWNLL'
           'TVTPIKTPDGRVSKFVGVQVDVTSKTEGKALA')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--members', type=int, default=10)
    ap.add_argument('--epochs', type=int, default=1000)      # configs/problem/protein_classifier_discrete.yaml
    ap.add_argument('--init', type=int, default=500)
    ap.add_argument('--per-round', type=int, default=100)
    ap.add_argument('--draw', type=int, default=10)
    ap.add_argument('--temperature', type=float, default=0.064)
    ap.add_argument('--stack', type=int, default=10, help='Thompson draws stacked per sampler call (1 = one draw at a time)')
    a = ap.parse_args()

    import sgpo_amd
    sgpo_amd.install_dropin()
    from sgpo_amd import runtime as rt, weights as W
    from sgpo_amd.tokenizer import Tokenizer
    from dataset.protein import ProteinPredictorDataset
    from models.mdlm import MDLM
    from problem.protein import ProteinPredictor
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    from training.train_classifier import train_classifier_ensemble
    dev = 'cuda:0'
    rt.init(dev)
    Lq = len(CREILOV)
    cfg = dict(diffusion='absorbing_state', parameterization='subs', time_conditioning=False,
               model=dict(hidden_size=512, cond_dim=128, n_blocks=8, n_heads=8, length=1024))
    net = MDLM('synthetic', cfg, None, num_steps=128, seq_len=Lq, device=dev, state_dict=W.dit_synthetic_state(seed=0))
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    np.random.seed(0)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, 'models'))
        for m, sd in enumerate(W.oracle_ensemble_synthetic_state(seed=5, n_members=10, n_res=Lq, hidden=400)):
            torch.save(sd, os.path.join(d, 'models', f'model{m}.pth'))
        os.makedirs(os.path.join(d, 'data', 'CreiLOV'))
        open(os.path.join(d, 'data', 'CreiLOV', 'parent.fasta'), 'w').write('>p\n' + CREILOV + '\n')
        cwd = os.getcwd()
        os.chdir(d)
        try:
            dc = dict(name='CreiLOV', data_path=None, full_seq=CREILOV, seq_len=Lq, alphabet=ALPHABET, alphabet_size=31,
                      residues=None, oracle_model_path=os.path.join(d, 'models'))
            ds = ProteinPredictorDataset(dc, Tokenizer(), capacity=1 << 16, max_batch=4096)
            # round 0: random variants of the parent (up to 3 substitutions)
            x0 = np.tile([ALPHABET.index(c) for c in CREILOV], (2 * a.init, 1))
            pos = rng.integers(0, Lq, (2 * a.init, 3))
            x0[np.arange(2 * a.init)[:, None], pos] = rng.integers(0, 20, (2 * a.init, 3))
            n0 = ds.update_data(torch.from_numpy(x0).to(dev), a.init, BO=False, round=0)
            torch.cuda.synchronize()

            # ---- train the ensemble (iterativeBO.py:143-179)
            t0 = time.perf_counter()
            X, Y = ds.tokens.cpu(), torch.as_tensor(ds.y, dtype=torch.float32).cpu()
            loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=64, shuffle=True)
            tc = dict(learning_rate=1e-3, wandb=False, n_epochs=a.epochs, n_epochs_no_time=a.epochs, batch_size=64)
            clfs = []
            for k in range(a.members):
                torch.manual_seed(42 + 10 + k)                 # config.seed + round * n_ensemble + idx
                clfs.append(ProteinPredictor(dc, dict(hidden_dim=64), device=dev))
            algo0 = Classifier_Guidance_Inpaint(net, dc, forward_op=clfs[0], temperature=a.temperature, device=dev)
            train_classifier_ensemble(clfs, net, loader, tc, save_dir=None, project_fn=algo0.project, time_conditioned=True)
            torch.cuda.synchronize()
            t_train = time.perf_counter() - t0
            with torch.no_grad():
                fit_mse = float(np.mean([float(((c.eval().log_likelihood(X.to(dev)) - Y.to(dev)) ** 2).mean()) for c in clfs]))

            # ---- Thompson sampling until the round is full (iterativeBO.py:196-231)
            t0 = time.perf_counter()
            n_new = it = 0
            max_it = max(5 * a.per_round // a.draw, 5)
            t_inf = t_upd = 0.0
            if a.stack > 1:
                # the wave form of the same loop (sgpo_amd.dist.thompson_round): `stack` draws in one sampler call, consumed
                # in draw order, so the round ends at the draw the loop below would end at
                from sgpo_amd.dist import thompson_round
                algo = Classifier_Guidance_Inpaint(net, dc, forward_op=clfs[0], temperature=a.temperature, device=dev)
                clock = dict(inf=0.0, upd=0.0)

                def draw_many(ms, keys):
                    t1 = time.perf_counter()
                    xs = algo.inference_many([clfs[m] for m in ms], keys, num_samples=a.draw)
                    torch.cuda.synchronize()
                    clock['inf'] += time.perf_counter() - t1
                    return [(x, None) for x in xs]

                def consume(t, _s):
                    t1 = time.perf_counter()
                    got = ds.update_data(algo.project_tokens(t.to(torch.int64)), a.per_round - consume.n, BO=True, round=1)
                    torch.cuda.synchronize()
                    clock['upd'] += time.perf_counter() - t1
                    consume.n += got
                    return got
                consume.n = 0
                n_new, it = thompson_round(None, consume, a.per_round, max_it, lambda: int(np.random.randint(0, a.members)), seed=7,
                                           rank=0, world=1, per_rank=a.stack, draw_many=draw_many)
                t_inf, t_upd = clock['inf'], clock['upd']
            while a.stack <= 1 and n_new < a.per_round and it < max_it:
                clf = clfs[np.random.randint(0, a.members)]
                algo = Classifier_Guidance_Inpaint(net, dc, forward_op=clf, temperature=a.temperature, device=dev)
                algo.update_model(clf)
                t1 = time.perf_counter()
                x = algo.inference(num_samples=a.draw)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                n_new += ds.update_data(algo.project_tokens(x), a.per_round - n_new, BO=True, round=1)
                torch.cuda.synchronize()
                t_inf += t2 - t1
                t_upd += time.perf_counter() - t2
                it += 1
            t_sample = time.perf_counter() - t0
            y = torch.as_tensor(ds.y, dtype=torch.float32)
            out = dict(metric='seconds per Bayesian-optimisation round (config 5, one GPU)', value=round(t_train + t_sample, 3),
                       unit='s', n_gpus=1, dtype='bf16 denoiser / fp32 value functions', data='synthetic',
                       config=dict(workload=f'MDLM (DiT 512 x 8) + cls_guidance_discrete on CreiLOV, {a.members} value-function '
                                            f'members x {a.epochs} epochs on {n0} sequences (batch 64, time-conditioned), then '
                                            f'Thompson draws of {a.draw} until {a.per_round} new sequences'),
                       train_s=round(t_train, 3), optimiser_steps=a.epochs * ((n0 + 63) // 64),
                       ms_per_lockstep_step=round(1e3 * t_train / (a.epochs * ((n0 + 63) // 64)), 4), train_fit_mse=round(fit_mse, 5),
                       sample_s=round(t_sample, 3), draws=it, new_sequences=int(n_new), inference_ms_per_draw=round(1e3 * t_inf / it, 2),
                       score_and_update_ms_per_draw=round(1e3 * t_upd / it, 2),
                       fitness_round0_mean=round(float(y[:n0].mean()), 4), fitness_round1_mean=round(float(y[n0:].mean()), 4))
        finally:
            os.chdir(cwd)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
