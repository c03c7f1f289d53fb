"""Benchmark of the SGPO guided-sampling hot path on MI355X.

Workload (BASELINE.json configs[1]): MDLM (DiT h512 / 8 blocks / 8 heads / cond 128 / K 31)
+ cls_guidance_discrete (exact guidance, temperature 0.064, MLP value function hidden 64) on
CreiLOV (L = 119), batch 4096 per GPU, 127 reverse steps.  One bench "step" = one full
`Classifier_Guidance_Inpaint.inference(num_samples=4096)` = 4096 guided sequences per GPU.
Weights are seeded synthetic (no checkpoints offline), tokens start fully masked, noise is
in-kernel Philox.  With N GPUs every rank samples its own 4096-sequence shard (weak scaling,
Philox counters offset by the global row) and the ranks all-gather tokens (uint8) + value
function scores over RCCL, as SURVEY.md section 8(e) lays out.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`--config {1,2,3,4,5}` selects another BASELINE configuration for the same JSON line (2 is the default and the
driver's headline): 3 = D3PM + daps_discrete on TrpB, 2048 sequences per GPU (a bench step is ONE of the run's 50
identical annealing steps, `value` is scaled to whole runs); 4 = MDLM + NOS_discrete on TrpB, one (strength, repeat)
cell of 200 sequences per GPU and step; 5 = Thompson draws of 10 sequences, one ensemble member per GPU and step.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed on the launch
stream during the timed steps) and `cpu_baseline` (the CPU oracle on a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CREILOV = ('MAGLRHTFVVADATLPDCPLVYASEGFYAMTGYGPDEVLGHNARFLQGEGTDPKEVQKIRDAIKKGEACSVRLLNYRKDGTPFWNLL'
           'TVTPIKTPDGRVSKFVGVQVDVTSKTEGKALA')
ALPHABET = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'
T_STEPS = 128
TEMPERATURE = 0.064
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X dense bf16 (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def build(device, batch, dit_seed=0, mlp_seed=1):
    import sgpo_amd
    sgpo_amd.install_dropin()
    from models.mdlm import MDLM
    from problem.protein import ProteinPredictor
    from sampling.cls_guidance import Classifier_Guidance_Inpaint
    from sgpo_amd import weights as W
    cfg = dict(diffusion='absorbing_state', parameterization='subs', time_conditioning=False,
               model=dict(hidden_size=512, cond_dim=128, n_blocks=8, n_heads=8, length=1024))
    sd = W.dit_synthetic_state(seed=dit_seed)
    net = MDLM('synthetic', cfg, None, num_steps=T_STEPS, seq_len=119, device=device, state_dict=sd)
    dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=119, full_seq=CREILOV, residues=None)
    pred = ProteinPredictor(dc, dict(hidden_dim=64), device=device)
    msd = W.mlp_synthetic_state(seed=mlp_seed, n_residues=119)
    pred.model.load_state_dict(msd)
    pred.eval()
    algo = Classifier_Guidance_Inpaint(net, dc, forward_op=pred, temperature=TEMPERATURE, device=device)
    return net, pred, algo, sd, msd


def cpu_baseline(sd, msd, n_seq=8, n_steps=48):
    """The CPU oracle (reference formulation incl. materialised jumps) on a bounded sample:
    n_seq sequences x the first n_steps of the 127 reverse steps, scaled to whole sequences."""
    from oracle import cls_guidance as ocg
    from oracle import mdlm as omdlm
    from oracle import mlp as omlp
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    pred = omlp.Predictor(msd)
    gen = torch.Generator().manual_seed(1234)
    x = 28 * torch.ones((n_seq, 119), dtype=torch.int64)
    ts = torch.linspace(T_STEPS - 1, 1, T_STEPS - 1, dtype=int)[:n_steps]
    t0 = time.perf_counter()
    with torch.no_grad():
        for t in ts:
            q = omdlm.p_sample(sd, x, t, T_STEPS, 28, hard=False)
            e = torch.empty(n_seq, 119, 31).exponential_(1, generator=gen)
            xx = x
            x, _ = ocg.cg_step(q, x, lambda qm: omdlm.get_guided_rates(pred, xx, t, qm, 31, TEMPERATURE)[0], e)
    dt = time.perf_counter() - t0
    per_seq = dt / n_seq * (T_STEPS - 1) / n_steps
    return dict(value=1.0 / per_seq, unit='sequences/s', cores=cores, kind='port',
                sample=f'{n_seq} sequences x {n_steps} of 127 guided steps (B={n_seq}), torch-CPU fp32 oracle, '
                       f'{dt:.1f} s measured, scaled linearly to 127 steps')


def pmc_traffic(kernel, batch):
    import glob
    if batch != 4096:        # the PMC passes were taken at the default bench batch
        return None
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*', '*_pmc_traffic.json')))
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if kernel in d:
            e = d[kernel]
            return dict(hbm_bytes_per_launch=e['hbm_bytes'], fetch_bytes=e['fetch_bytes'], write_bytes=e['write_bytes'],
                        l2_hit=round(e['l2_hit'], 3), source=os.path.relpath(f, ROOT))
    return None


def gemm_algorithmic_bytes(kind, batch, L=119, h=512, ff=2048):
    """Algorithmic HBM bytes of one launch of a fused denoiser GEMM: the bf16 A
    operand and weights once, the output once, and for GATE_RES the fp32 residual read and written.
    GATE_RES has two shapes per block (K = h and K = ff); the profiler averages them, so does this."""
    M = batch * L
    if 'QKV_ROPE' in kind:
        return M * h * 2 + 3 * h * h * 2 + M * 3 * h * 2
    if 'GATE_RES' in kind:
        a = (M * h * 2 + h * h * 2 + M * ff * 2 + h * ff * 2) / 2
        ln = M * h * 2 if 'GATE_RES_LN' in kind else 0          # fused kernel also writes the bf16 LayerNorm-modulate rows
        return a + 2 * M * h * 4 + batch * h * 4 + ln
    if 'BIAS_GELU' in kind:
        return M * h * 2 + h * ff * 2 + M * ff * 2
    return None

# ---------------------------------------------------------------------------------------------------------------
# The other BASELINE configurations through the same contract (`--config 3|4|5`).  Config 2 above stays the default.
TRPB_RES = [117, 118, 119, 162, 166, 182, 183, 184, 185, 186, 227, 228, 230, 231, 301]
L_TRPB = 389


def _trpb_parent():
    g = torch.Generator().manual_seed(0)
    return ''.join(ALPHABET[int(t)] for t in torch.randint(0, 20, (L_TRPB,), generator=g))


class _Config3:
    """D3PM (ByteNet 16 x d1024) + daps_discrete on TrpB; the batch shards over ranks (2048 sequences per GPU).  A run is
    50 identical annealing steps (19 denoiser forwards + posteriors, 1000 MH proposals, q_sample); ONE of them is a bench
    step and `value` credits batch / 50 sequences per step."""
    default_batch, steps_per_run = 2048, 50

    def __init__(self, device, batch, rank, world):
        import sgpo_amd
        sgpo_amd.install_dropin()
        from models.d3pm import D3PM
        from problem.protein import ProteinPredictor
        from sampling.daps import DAPS
        from sgpo_amd import weights as W
        self.B, self.rank, self.world = batch, rank, world
        self.bn_sd = W.bytenet_synthetic_state(seed=0)
        self.net = D3PM('synthetic', L_TRPB, device=device, state_dict=self.bn_sd)
        dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=L_TRPB, full_seq=_trpb_parent(), residues=TRPB_RES)
        self.pred = ProteinPredictor(dc, dict(hidden_dim=64), device=device)
        self.msd = W.mlp_synthetic_state(seed=1, n_residues=len(TRPB_RES))
        self.pred.model.load_state_dict(self.msd)
        self.pred.eval()
        self.algo = DAPS(self.net, forward_op=self.pred, num_steps=50, ode_steps=20, mh_steps=1000, alpha=128., beta=1000,
                         data_config=dc, device=device)
        self.xt = None

    def step(self, i):
        """Annealing step i (mod 49, so that q_sample is always part of it) of the 50-step schedule on the running state."""
        a, net = self.algo, self.net
        if self.xt is None or i < 0:
            net.reseed(seed=7, row_offset=self.rank * self.B)
            self.xt = net.get_start(self.B)
        k = i % (a.num_steps - 1)
        t_now = a.time_steps[k]
        x0 = a.uncond_sample(net, self.xt, t_now, False)
        x0 = a.metropolis_hasting(x0, a.forward_op, None, a._sigma_at(t_now), steps=a.mh_steps, alpha=a.alpha)
        self.xt = net.q_sample(x0, a.time_steps[k + 1].repeat(self.B))
        return x0, self.pred.log_likelihood(x0)

    def units(self):
        return self.world * self.B / self.steps_per_run

    def n_launches(self, steps):
        return steps * 3000 + 64

    def describe(self):
        return {'workload': 'D3PM (ByteNet 16 x d1024/h512, k5) + daps_discrete on TrpB, L=389, 15 residues, %d sequences per GPU; '
                            'one bench step = one of the 50 annealing steps of a run (19 denoiser forwards + fp64 posteriors, 1000 MH '
                            'proposals, q_sample); value = sequences per second of whole 50-step runs at that step time' % self.B,
                'denoiser': 'ByteNetLMTime restated (third party, parity unpinned), bf16 MFMA / fp32 accumulate',
                'global_batch': self.world * self.B, 'parallelism': f'dp{self.world} (batch rows sharded + all-gather of tokens/scores)'}

    def cpu_baseline(self):
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import bench_daps
        return bench_daps.cpu_baseline(self.bn_sd, self.msd)


class _Config4:
    """MDLM + NOS_discrete on TrpB (pareto.py sweep): one (guidance strength, repeat) cell of 200 sequences per GPU and
    bench step; cells are dealt to the ranks and gathered once (sgpo_amd.dist.run_cells)."""
    default_batch = 200
    COEFS = [0.1, 1.0, 10.0, 100.0, 200.0]

    def __init__(self, device, batch, rank, world):
        import sgpo_amd
        sgpo_amd.install_dropin()
        from models.mdlm import MDLM
        from problem.protein_nos import NOSHead
        from sampling.NOS import NOS
        from sgpo_amd import weights as W
        self.B, self.rank, self.world, self.device = batch, rank, world, device
        cfg = dict(diffusion='absorbing_state', parameterization='subs', time_conditioning=False,
                   model=dict(hidden_size=512, cond_dim=128, n_blocks=8, n_heads=8, length=1024))
        self.sd = W.dit_synthetic_state(seed=0)
        self.net = MDLM('synthetic', cfg, None, num_steps=T_STEPS, seq_len=L_TRPB, device=device, state_dict=self.sd)
        dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=L_TRPB, full_seq=_trpb_parent(), residues=TRPB_RES)
        torch.manual_seed(1)
        self.head = NOSHead(512).to(device).eval()
        self.algo = NOS(self.net, dc, forward_op=self.head, num_nos_steps=5, nos_step_size=0.5, nos_stability_coef=0.1, device=device)

    def _cell(self, cell, key):
        coef, _rep = cell
        self.algo.nos_stability_coef = coef
        self.net.reseed(seed=key, row_offset=0)
        x = self.algo.inference(num_samples=self.B, verbose=False)
        return x, None

    def step(self, i):
        from sgpo_amd.dist import run_cells
        cells = [(self.COEFS[(i * self.world + r) % len(self.COEFS)], (i * self.world + r) // len(self.COEFS)) for r in range(self.world)]
        out = run_cells(cells, self._cell, seed=42 + i, rank=self.rank, world=self.world, rows_per_cell=self.B)
        return out[0][0], None

    def units(self):
        return self.world * self.B

    def n_launches(self, steps):
        return steps * (T_STEPS - 1) * 80 + 64

    def describe(self):
        return {'workload': 'MDLM + NOS_discrete on TrpB, L=389, 127 reverse steps x (1 denoiser forward + 5 hidden-state optimisation '
                            'steps of size 0.5), %d sequences per (guidance strength, repeat) cell, one cell per GPU and bench step, '
                            'strengths %s' % (self.B, self.COEFS),
                'denoiser': 'DiT h512 x 8 blocks x 8 heads, cond 128, bf16 MFMA / fp32 accumulate', 'global_batch': self.world * self.B,
                'parallelism': f'{self.world} cells at a time (independent sampler calls + one all-gather of tokens)'}

    def cpu_baseline(self, n_seq=8, n_steps=12):
        from oracle import nos as onos
        cores = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(cores)
        g = torch.Generator().manual_seed(5)
        x = torch.full((n_seq, L_TRPB), 28, dtype=torch.int64)
        w, b = torch.randn(1, 512, generator=g) * 0.5, torch.randn(1, generator=g)
        t0 = time.perf_counter()
        for k in range(n_steps):
            u = torch.rand(n_seq, L_TRPB, 31, generator=g)
            x, _, _ = onos.nos_denoise(self.sd, w, b, x, T_STEPS - 1 - k, T_STEPS, 28, 5, 0.5, 0.1, u)
            x = x.clone()
            x[:, ::2] = 28                  # keep half the positions masked so every step does the masked-row work
        dt = time.perf_counter() - t0
        per_seq = dt / n_seq * (T_STEPS - 1) / n_steps
        return dict(value=1.0 / per_seq, unit='sequences/s', cores=cores, kind='port',
                    sample=f'{n_seq} sequences x {n_steps} of 127 NOS reverse steps (autograd + Adagrad oracle, torch-CPU fp32), '
                           f'{dt:.1f} s measured, scaled linearly to 127 steps')


class _Config5(_Config4):
    """iterativeBO.py Thompson sampling: every GPU draws an ensemble member and samples 10 guided sequences with it
    (MDLM + cls_guidance_discrete on CreiLOV); a bench step is one wave of `world` draws, gathered once."""
    default_batch = 10

    def __init__(self, device, batch, rank, world):
        self.B, self.rank, self.world = batch, rank, world
        self.net, pred0, self.algo, self.sd, self.msd = build(device, batch)
        from problem.protein import ProteinPredictor
        from sgpo_amd import weights as W
        dc = dict(alphabet=ALPHABET, alphabet_size=31, seq_len=119, full_seq=CREILOV, residues=None)
        self.members = []
        for m in range(10):
            p = ProteinPredictor(dc, dict(hidden_dim=64), device=device)
            p.model.load_state_dict(W.mlp_synthetic_state(seed=100 + m, n_residues=119))
            self.members.append(p.eval())
        import numpy as np
        self.rng = np.random.RandomState(42)

    def _draw(self, member, key):
        clf = self.members[member]
        self.algo.update_model(clf)
        self.net.reseed(seed=key, row_offset=0)
        x = self.algo.inference(num_samples=self.B)
        return x, clf.log_likelihood(x)

    DRAWS = 10      # Thompson draws stacked per GPU and bench step (one denoiser forward per reverse step for all of them)

    def _draw_many(self, members, keys):
        xs = self.algo.inference_many([self.members[m] for m in members], keys, num_samples=self.B)
        return [(x, self.members[m].log_likelihood(x)) for m, x in zip(members, xs)]

    def step(self, i):
        from sgpo_amd.dist import thompson_round
        got = []
        thompson_round(self._draw, lambda t, s: (got.append((t, s)) or t.shape[0]), n_needed=10 ** 9,
                       max_draws=self.world * self.DRAWS, pick_member=lambda: int(self.rng.randint(0, 10)), seed=1000 + i,
                       rank=self.rank, world=self.world, per_rank=self.DRAWS, draw_many=self._draw_many, rows_per_draw=self.B)
        return torch.cat([t for t, _ in got]), torch.cat([s for _, s in got])

    def units(self):
        return self.world * self.B * self.DRAWS

    def n_launches(self, steps):
        return steps * (T_STEPS - 1) * (80 + 4 * self.DRAWS) + 64

    def describe(self):
        return {'workload': 'MDLM + cls_guidance_discrete (temperature 0.064) on CreiLOV, L=119, 127 reverse steps, Thompson draws of '
                            '%d sequences, each guided by one of 10 value-function members; %d draws per GPU and bench step, stacked '
                            '(one denoiser forward per reverse step for all of them, guidance + reverse step per draw; tokens '
                            'identical to one draw at a time; ensemble training is outside the timed region: tools/bench_bo.py '
                            'times a whole round)' % (self.B, self.DRAWS),
                'denoiser': 'DiT h512 x 8 blocks x 8 heads, cond 128, bf16 MFMA / fp32 accumulate', 'global_batch': self.world * self.B,
                'parallelism': f'{self.world} x {self.DRAWS} draws at a time (independent sampler calls + one all-gather of tokens/scores)'}

    def cpu_baseline(self):
        return cpu_baseline(self.sd, self.msd, n_seq=10, n_steps=24)


class _Config1(_Config5):
    """prior_sample.py: unguided prior samples of the MDLM denoiser on CreiLOV (temperature 0), 64 sequences per call."""
    default_batch = 64

    def __init__(self, device, batch, rank, world):
        self.B, self.rank, self.world = batch, rank, world
        self.net, self.pred, self.algo, self.sd, self.msd = build(device, batch)
        self.algo.temperature = 0.0

    def step(self, i):
        self.net.reseed(seed=2000 + i, row_offset=self.rank * self.B)
        x = self.algo.inference(num_samples=self.B)
        return x, self.pred.log_likelihood(x)

    def units(self):
        return self.world * self.B        # (not the parent's: no stacked draws here)

    def n_launches(self, steps):
        return steps * (T_STEPS - 1) * 80 + 64

    def describe(self):
        return {'workload': 'MDLM prior sampling (cls_guidance_discrete at temperature 0) on CreiLOV, L=119, 127 reverse steps, '
                            '%d sequences per GPU and bench step' % self.B,
                'denoiser': 'DiT h512 x 8 blocks x 8 heads, cond 128, bf16 MFMA / fp32 accumulate', 'global_batch': self.world * self.B,
                'parallelism': f'dp{self.world} (independent shards + all-gather of tokens/scores)'}

    def cpu_baseline(self):
        """Unguided steps of the CPU restatement (one denoiser forward + reverse step each) on a bounded sample."""
        from oracle import cls_guidance as ocg
        from oracle import mdlm as omdlm
        cores = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(cores)
        n_seq, n_steps = 64, 12
        gen = torch.Generator().manual_seed(7)
        x = 28 * torch.ones((n_seq, 119), dtype=torch.int64)
        ts = torch.linspace(T_STEPS - 1, 1, T_STEPS - 1, dtype=int)[:n_steps]
        t0 = time.perf_counter()
        with torch.no_grad():
            for t in ts:
                q = omdlm.p_sample(self.sd, x, t, T_STEPS, 28, hard=False)
                e = torch.empty(n_seq, 119, 31).exponential_(1, generator=gen)
                x, _ = ocg.cg_step(q, x, lambda qm: qm, e)
        dt = time.perf_counter() - t0
        per_seq = dt / n_seq * (T_STEPS - 1) / n_steps
        return dict(value=1.0 / per_seq, unit='sequences/s', cores=cores, kind='port',
                    sample=f'{n_seq} sequences x {n_steps} of 127 unguided steps, torch-CPU fp32 oracle, {dt:.1f} s measured, '
                           'scaled linearly to 127 steps')


def _dist_init(world, local_rank):
    """One process per GPU over RCCL (`nccl`).  SGPO_BENCH_BACKEND=gloo is the rehearsal of the N > 1 control flow on a box
    with fewer GPUs than ranks (RCCL refuses two ranks on one device): ranks share the GPUs round-robin and the gathers are
    staged through the host; it is never used for a reported number."""
    backend = os.environ.get('SGPO_BENCH_BACKEND', 'nccl')
    index = local_rank if backend == 'nccl' else local_rank % max(torch.cuda.device_count(), 1)
    device = f'cuda:{index}'
    torch.cuda.set_device(index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(device))
        else:
            dist.init_process_group(backend)
    return dist, device, backend


def _max_over_ranks(dist, backend, dt, device):
    tt = torch.tensor([dt], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def main_other(a):
    """Configs 1 / 3 / 4 / 5 through the bench contract: W warm-up steps, K timed steps between barrier + synchronize, max over
    ranks, one JSON line from rank 0 with `roofline` (dominant kernel, HIP events on the launch stream) and `cpu_baseline`."""
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1 and a.gpus > 1:
        raise SystemExit('launch with torch.distributed.run --nproc-per-node N for --gpus N')
    dist, device, backend = _dist_init(world, local_rank)
    line = run_other(a, dist, device, backend, rank, world)
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_other(a, dist, device, backend, rank, world):
    """One of configs 1 / 3 / 4 / 5 (a.config) for a.warmup + a.steps steps in THIS process; the JSON line as a dict (rank 0)."""
    from sgpo_amd import _lib as L
    cls = {1: _Config1, 3: _Config3, 4: _Config4, 5: _Config5}[a.config]
    wl = cls(device, a.batch or cls.default_batch, rank, world)
    lib = L.load()

    def one_step(i):
        x, sc = wl.step(i)
        if world > 1 and a.config in (1, 3):     # batch rows sharded: ONE all-gather of tokens (uint8) + scores
            from sgpo_amd.dist import gather_samples
            return gather_samples(x, sc.to(torch.float32), max_rows=x.shape[0])
        return x, sc

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        one_step(-1 - i)
    sync()
    # Small batches (configs 1 and 5: kernels of 5-30 us) are launch-bound, and two event records per kernel cost 40-60 % of
    # the wall clock there (config 5: 100 ms per draw without, 162 ms with them).  Their K steps are timed WITHOUT the
    # per-kernel events; the kernel table and the roofline come from a second pass over the same K steps right after.
    prof_in_timed_region = a.config not in (1, 5)
    if not a.no_prof and prof_in_timed_region:
        L.check(lib.sgpo_prof_enable(1, wl.n_launches(a.steps)), 'prof_enable')
    t0 = time.perf_counter()
    for i in range(a.steps):
        out = one_step(i)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = _max_over_ranks(dist, backend, dt, device)
    dt_prof = dt
    if not a.no_prof and not prof_in_timed_region:
        graphs = getattr(wl.net, 'use_graphs', False)
        wl.net.use_graphs = False        # the per-kernel events are recorded by eager launches, not by graph replays
        L.check(lib.sgpo_prof_enable(1, wl.n_launches(a.steps)), 'prof_enable')
        t1 = time.perf_counter()
        for i in range(a.steps):
            one_step(i)
        sync()
        dt_prof = time.perf_counter() - t1
    roof, kernels = None, {}
    if not a.no_prof:
        n = L.PROF_KINDS
        ms, cnt, work = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)()
        L.check(lib.sgpo_prof_read(n, ms, cnt, work), 'prof_read')
        lib.sgpo_prof_enable(0, 0)
        tot = sum(ms)
        for k in range(n):
            if cnt[k]:
                kernels[lib.sgpo_prof_kind_name(k).decode()] = dict(total_ms=round(ms[k], 3), launches=int(cnt[k]),
                                                                    avg_us=round(1e3 * ms[k] / cnt[k], 2), share=round(ms[k] / tot, 4),
                                                                    work_per_launch=work[k] / cnt[k])
        dom = max(range(n), key=lambda k: ms[k])
        name = lib.sgpo_prof_kind_name(dom).decode()
        per_launch, avg_s = work[dom] / cnt[dom], ms[dom] / cnt[dom] * 1e-3
        if name.startswith(('gemm', 'attention')):      # work = flops (2 M N K); the fused GEMMs of these shapes are MFMA-side
            ach = per_launch / avg_s / 1e12
            roof = dict(bound='mfma', kernel=name, achieved=round(ach, 2), peak=MFMA_BF16_PEAK_TFLOPS, unit='TFLOP/s',
                        frac=round(ach / MFMA_BF16_PEAK_TFLOPS, 4), traffic=None, avg_launch_us=round(avg_s * 1e6, 2),
                        flops_per_launch=per_launch)
        else:                                           # work = algorithmic bytes
            ach = per_launch / avg_s / 1e9
            roof = dict(bound='hbm', kernel=name, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=round(ach / HBM_PEAK_GBS, 4), traffic=None, avg_launch_us=round(avg_s * 1e6, 2),
                        bytes_per_launch=per_launch)
        gf = sum(work[k] for k in range(n) if lib.sgpo_prof_kind_name(k).decode().startswith(('gemm', 'attention')))
        gt = sum(ms[k] for k in range(n) if lib.sgpo_prof_kind_name(k).decode().startswith(('gemm', 'attention')))
        if gt > 0:
            roof['denoiser_tflops'] = round(gf / (gt * 1e-3) / 1e12, 2)
        roof['gpu_busy_frac'] = round(tot * 1e-3 / dt_prof, 4)
        if not prof_in_timed_region:
            wl.net.use_graphs = graphs
            roof['note'] = ('kernel durations from a second pass over the same %d steps with per-kernel events on (%.1f ms per step '
                            'there): at this batch size the event records themselves are 40-60 %% of the wall clock, so the timed '
                            'region runs without them; gpu_busy_frac = kernel time of that pass / wall clock of the TIMED region '
                            '(about 1: the small batches are bound by the per-kernel fixed cost on the GPU, not by launch gaps -- '
                            'replaying the loop from a recorded HIP graph, MDLM.use_graphs, changes the wall clock by 1-2 %%)'
                            % (a.steps, 1e3 * dt_prof / a.steps))
            roof['gpu_busy_frac'] = round(tot * 1e-3 / dt, 4)
    if rank != 0:
        return None
    cpu = wl.cpu_baseline() if (not a.no_cpu_baseline and world == 1) else None
    cfg = wl.describe()
    cfg['baseline_config'] = a.config
    return {
        'metric': 'guided sequences/sec (whole node)', 'value': round(wl.units() * a.steps / dt, 3), 'unit': 'sequences/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(1e3 * dt / a.steps, 2),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic', 'config': cfg,
        'roofline': roof, 'cpu_baseline': cpu, 'kernels': kernels, 'checksum': int(out[0].long().sum())}


def config_legs(a, dist, device, backend, rank, world):
    """After the headline region of the default run: short legs of BASELINE configs 1, 3, 4 and 5 in the same process
    (the `_Config*` workloads of `--config N`, 2 timed steps after 1 warm-up each), so that the driver's one JSON line holds
    a measured number for every configuration.  {'3': {value, unit, ms_per_step, kernel, frac, bound, cpu_baseline, ...}}."""
    import copy
    legs = {}
    for cid in (1, 3, 4, 5):
        b = copy.copy(a)
        b.config, b.batch, b.steps, b.warmup = cid, None, 2, 1
        try:
            line = run_other(b, dist, device, backend, rank, world)
        except Exception as exc:           # a leg must never take the headline line down with it
            legs[str(cid)] = {'error': f'{type(exc).__name__}: {exc}'}
            continue
        torch.cuda.empty_cache()
        if line is None:
            continue
        r = line['roofline'] or {}
        legs[str(cid)] = {'value': line['value'], 'unit': line['unit'], 'ms_per_step': line['ms_per_step'], 'steps': b.steps,
                          'warmup': b.warmup, 'workload': line['config']['workload'], 'kernel': r.get('kernel'),
                          'bound': r.get('bound'), 'achieved': r.get('achieved'), 'roofline_unit': r.get('unit'),
                          'frac': r.get('frac'), 'avg_launch_us': r.get('avg_launch_us'),
                          'denoiser_tflops': r.get('denoiser_tflops'), 'gpu_busy_frac': r.get('gpu_busy_frac'),
                          'cpu_baseline': line['cpu_baseline'], 'checksum': line['checksum']}
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', type=int, default=2, choices=[1, 2, 3, 4, 5], help='BASELINE configuration (2 = headline)')
    ap.add_argument('--batch', type=int, default=None, help='sequences per GPU per step (default: the configuration\'s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prof', action='store_true')
    ap.add_argument('--no-skip-leg', action='store_true', help='skip the second (skip_unmasked) measurement')
    ap.add_argument('--no-legs', action='store_true', help='default run only: skip the short legs of configs 1, 3, 4, 5')
    a = ap.parse_args()
    if a.config != 2:
        return main_other(a)
    a.batch = a.batch or 4096

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node N for --gpus N')
    dist, device, backend = _dist_init(world, local_rank)

    from sgpo_amd import _lib as L
    net, pred, algo, sd, msd = build(device, a.batch)
    lib = L.load()
    B = a.batch

    def one_step(step_idx):
        # shard-invariant noise: global row offset of this rank's shard; new key per step
        net.reseed(seed=42 + step_idx, row_offset=rank * B)
        x = algo.inference(num_samples=B)
        score = pred.log_likelihood(x)
        if world > 1:      # ONE all-gather of tokens (uint8) + scores per batch (sgpo_amd.dist.gather_rows)
            from sgpo_amd.dist import gather_samples
            return gather_samples(x, score, max_rows=B)
        return x, score

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        one_step(-1 - i)
    sync()
    n_launch = a.steps * (T_STEPS - 1) * 80 + 64
    if not a.no_prof:
        L.check(lib.sgpo_prof_enable(1, n_launch), 'prof_enable')
    t0 = time.perf_counter()
    out = None
    for i in range(a.steps):
        out = one_step(i)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = _max_over_ranks(dist, backend, dt, device)

    roof = None
    kernels = {}
    if not a.no_prof:
        L.check(lib.sgpo_prof_enable(0, 0), 'prof_pause')       # stop recording, keep the records (read below)
    # SECOND number, never the headline: the same steps with `skip_unmasked` -- sequences that hold no mask skip the
    # denoiser forward of that step (output-identical, checked here by the token checksum of the same seeds).
    skip = None
    if not a.no_skip_leg:
        k_skip = min(a.steps, 3)
        want = [None] * k_skip
        net.skip_unmasked = True
        one_step(-1)
        net.skipped_rows = net.denoised_rows = 0
        sync()
        t1 = time.perf_counter()
        for i in range(k_skip):
            want[i] = one_step(i)[0]
        sync()
        dts = time.perf_counter() - t1
        if world > 1:
            dts = _max_over_ranks(dist, backend, dts, device)
        net.skip_unmasked = False
        same = bool(torch.equal(want[k_skip - 1], out[0])) if k_skip == a.steps else None
        skip = dict(value=round(world * B * k_skip / dts, 2), unit='sequences/s', steps=k_skip, ms_per_step=round(1e3 * dts / k_skip, 2),
                    rows_skipped_frac=round(net.skipped_rows / max(net.skipped_rows + net.denoised_rows, 1), 4),
                    tokens_identical_to_headline_run=same,
                    note='opt-in MDLM(skip_unmasked=True): the denoiser runs only on sequences that still hold a mask; the '
                         'headline `value` keeps the reference\'s one full forward per step')
    if not a.no_prof:
        n = L.PROF_KINDS
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        work = (C.c_double * n)()
        L.check(lib.sgpo_prof_read(n, ms, cnt, work), 'prof_read')
        lib.sgpo_prof_enable(0, 0)
        tot = sum(ms)
        for k in range(n):
            if cnt[k]:
                kernels[lib.sgpo_prof_kind_name(k).decode()] = dict(
                    total_ms=round(ms[k], 3), launches=int(cnt[k]), avg_us=round(1e3 * ms[k] / cnt[k], 2),
                    share=round(ms[k] / tot, 4), work_per_launch=work[k] / cnt[k])
        # every kernel against its own roof: the binding one of (flops / MFMA peak, algorithmic bytes / HBM peak)
        M_rows, Hd = B * 119, 512
        for kname, kd in kernels.items():
            t_s = kd['avg_us'] * 1e-6
            flops = nbytes = None
            if kname.startswith('gemm'):
                flops, nbytes = kd['work_per_launch'], gemm_algorithmic_bytes(kname, B)
                if nbytes is None and 'BIAS_F32' in kname:          # vocabulary head: bf16 rows in, 31 fp32 logits out
                    nbytes = M_rows * Hd * 2 + M_rows * 31 * 4
            elif kname.startswith('attention'):
                flops, nbytes = kd['work_per_launch'], M_rows * Hd * 2 * 4   # q, k, v read, output written (bf16)
            elif kname.startswith('mlp_jump'):
                kd['roof'] = dict(bound='fp32 mfma', achieved=round(kd['work_per_launch'] / t_s / 1e12, 1), peak=157.3,
                                  unit='TFLOP/s', frac=round(kd['work_per_launch'] / t_s / 1e12 / 157.3, 4))
                continue
            else:
                nbytes = kd['work_per_launch']
            t_m = flops / (MFMA_BF16_PEAK_TFLOPS * 1e12) if flops else 0.0
            t_h = nbytes / (HBM_PEAK_GBS * 1e9) if nbytes else 0.0
            if t_h >= t_m:
                kd['roof'] = dict(bound='hbm', achieved=round(nbytes / t_s / 1e9, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                                  frac=round(nbytes / t_s / 1e9 / HBM_PEAK_GBS, 4))
            else:
                kd['roof'] = dict(bound='mfma', achieved=round(flops / t_s / 1e12, 1), peak=MFMA_BF16_PEAK_TFLOPS, unit='TFLOP/s',
                                  frac=round(flops / t_s / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4))
        dom = max(range(n), key=lambda k: ms[k])
        name = lib.sgpo_prof_kind_name(dom).decode()
        is_mfma = name.startswith('gemm') or name.startswith('attention') or name.startswith('mlp_jump')
        per_launch = work[dom] / cnt[dom]
        avg_s = ms[dom] / cnt[dom] * 1e-3
        if is_mfma:
            # a fused GEMM moves its operands, its output and (GATE_RES) the fp32 residual both ways; at
            # K = 512 the two roofs are close, so both are priced and the binding one (larger minimum
            # time at peak) is the reported bound
            ach = per_launch / avg_s / 1e12
            ab = gemm_algorithmic_bytes(name, B)
            t_mfma = per_launch / (MFMA_BF16_PEAK_TFLOPS * 1e12)
            t_hbm = ab / (HBM_PEAK_GBS * 1e9) if ab else 0.0
            views = dict(mfma=dict(achieved_TFLOPs=round(ach, 2), frac=round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                                   min_us_at_peak=round(t_mfma * 1e6, 1)))
            if ab:
                views['hbm'] = dict(achieved_GBs=round(ab / avg_s / 1e9, 1), frac=round(ab / avg_s / 1e9 / HBM_PEAK_GBS, 4),
                                    min_us_at_peak=round(t_hbm * 1e6, 1), algorithmic_bytes_per_launch=ab)
            if t_hbm > t_mfma:
                roof = dict(bound='hbm', kernel=name, achieved=views['hbm']['achieved_GBs'], peak=HBM_PEAK_GBS, unit='GB/s',
                            frac=views['hbm']['frac'], traffic=None, avg_launch_us=round(avg_s * 1e6, 2),
                            bytes_per_launch=ab, flops_per_launch=per_launch, both_roofs=views)
            else:
                roof = dict(bound='mfma', kernel=name, achieved=round(ach, 2), peak=MFMA_BF16_PEAK_TFLOPS, unit='TFLOP/s',
                            frac=round(ach / MFMA_BF16_PEAK_TFLOPS, 4), traffic=None,
                            avg_launch_us=round(avg_s * 1e6, 2), flops_per_launch=per_launch, both_roofs=views)
        else:
            ach = per_launch / avg_s / 1e9
            roof = dict(bound='hbm', kernel=name, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=round(ach / HBM_PEAK_GBS, 4), traffic=None, avg_launch_us=round(avg_s * 1e6, 2),
                        bytes_per_launch=per_launch)
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the
        # per-launch figure comes from the committed rocprofv3 --pmc summary of the same kernel at the same
        # shape (profiles/<round>/*_pmc_traffic.json, written by tools/dbg/pmc_run.sh + pmc_agg.py); null if absent
        tr = pmc_traffic(name, B)
        roof['traffic'] = tr['hbm_bytes_per_launch'] if tr else None    # bytes per launch
        roof['traffic_detail'] = tr
        # whole-denoiser MFMA utilisation (all GEMM + attention flops over their summed time)
        gf = sum(work[k] for k in range(n) if lib.sgpo_prof_kind_name(k).decode().startswith(('gemm', 'attention')))
        gt = sum(ms[k] for k in range(n) if lib.sgpo_prof_kind_name(k).decode().startswith(('gemm', 'attention')))
        if gt > 0:
            roof['denoiser_tflops'] = round(gf / (gt * 1e-3) / 1e12, 2)
            roof['gpu_busy_frac'] = round(tot * 1e-3 / dt, 4)

    # row f-1 (the step right after the path): fitness-oracle ensemble on the sampled batch, timed
    # outside the headline region (the reference evaluates it in dataset.update_data, not in inference())
    fit = None
    if rank == 0:
        from sgpo_amd import runtime as rt
        from sgpo_amd import weights as W
        ens = rt.FitnessEnsemble(W.oracle_ensemble_synthetic_state(seed=2, n_members=10, n_res=119), device=device)
        xs = out[0][:B].to(torch.int64).clamp(max=19)          # tokens outside the 20 letters raise in the reference
        parent = torch.tensor([ALPHABET.index(c) for c in CREILOV], device=device)
        ens.forward(xs, parent, 70, 0.99)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ens.forward(xs, parent, 70, 0.99)
        e1.record()
        torch.cuda.synchronize()
        fms = e0.elapsed_time(e1) / 5
        fit = dict(workload='fitness-oracle ensemble, 10 members x (2380 -> 400 -> 1), %d sequences' % B,
                   ms=round(fms, 3), sequences_per_s=round(B / (fms * 1e-3), 1),
                   table_gather_GBps=round(B * 10 * 119 * 400 * 4 / (fms * 1e-3) / 1e9, 1))
    # row f-2 (between the sampler and the fitness oracle): token-level project_sequences + update_data's
    # de-duplication against a store that already holds the earlier batches; outside the headline region too
    acq = None
    if rank == 0:
        from sgpo_amd import runtime as rt
        xs = out[0][:B].to(torch.int64).contiguous()
        Lx = xs.shape[1]
        parent = torch.tensor([ALPHABET.index(c) for c in CREILOV], device=device)
        gen = torch.Generator(device=device).manual_seed(3)
        u = torch.rand(B, 11, device=device, generator=gen)
        keys = torch.rand(B, Lx, device=device, generator=gen)
        store = rt.AcquisitionSet(Lx, capacity=16 * B, max_batch=B, device=device)
        reps = 8
        batches = [torch.where(torch.rand(B, Lx, device=device, generator=gen) < 0.02,
                               torch.randint(0, 20, (B, Lx), device=device, generator=gen), xs) for _ in range(reps + 1)]
        rt.project_tokens(batches[0].clone(), parent, 20, 11, u, 10, keys)
        store.update(batches[0], B, 1, 0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        kept_total = 0
        for r in range(reps):
            y = rt.project_tokens(batches[r + 1], parent, 20, 11, u, 10, keys)
            kept, _ = store.update(y, B, 1, r + 1)
            kept_total += int(kept.numel())
        torch.cuda.synchronize()
        ams = (time.perf_counter() - t1) * 1e3 / reps
        acq = dict(workload='project_tokens (n_max_mutations 10) + acquisition-set update (BO scope), %d x %d tokens' % (B, Lx),
                   ms=round(ams, 3), sequences_per_s=round(B / (ams * 1e-3), 1), appended_per_call=kept_total / reps,
                   note='wall clock incl. the two host synchronisations of sgpo_acq_update')
    # row f-3: value-function ensemble training (10 members in lockstep, batch 64), outside the headline region
    trn = None
    if rank == 0:
        from sgpo_amd import runtime as rt
        from sgpo_amd import weights as W
        E_m, B_t, n_steps = 10, 64, 100
        trainer = rt.MlpEnsembleTrainer([W.mlp_synthetic_state(seed=50 + e, n_residues=119) for e in range(E_m)], list(range(119)),
                                        device=device)
        gen = torch.Generator(device=device).manual_seed(4)
        xtr = torch.randint(0, 31, (E_m, B_t, 119), device=device, generator=gen)
        ttr = torch.randint(0, T_STEPS, (E_m, B_t), device=device, generator=gen).float()
        ytr = torch.randn(E_m, B_t, device=device, generator=gen)
        for _ in range(5):
            trainer.step(xtr, ttr, ytr)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(n_steps):
            trainer.step(xtr, ttr, ytr)
        torch.cuda.synchronize()
        tms = (time.perf_counter() - t2) * 1e3 / n_steps
        trn = dict(workload='Adam/MSE step of MLPModel (119 residues, 510 k parameters), %d members in lockstep, batch %d' % (E_m, B_t),
                   ms_per_step=round(tms, 4), member_steps_per_s=round(E_m / (tms * 1e-3), 1))
    if rank == 0:
        cpu = None
        if not a.no_cpu_baseline and world == 1:      # reported at N = 1 only
            cpu = cpu_baseline(sd, msd)
        total_seq = world * B * a.steps
        line = {
            'metric': 'guided sequences/sec (whole node)', 'value': round(total_seq / dt, 2), 'unit': 'sequences/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(1e3 * dt / a.steps, 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'MDLM + cls_guidance_discrete (exact guidance, temperature 0.064) on CreiLOV, '
                                   'L=119, K=31, 127 reverse steps, batch %d per GPU' % B,
                       'denoiser': 'DiT h512 x 8 blocks x 8 heads, cond 128, bf16 MFMA / fp32 accumulate',
                       'value_function': 'MLPModel hidden 64, fp32', 'global_batch': world * B,
                       'parallelism': f'dp{world} (independent shards + all-gather of tokens/scores)'},
            'value_skip_unmasked': skip['value'] if skip else None, 'skip_unmasked': skip,
            'roofline': roof, 'cpu_baseline': cpu, 'fitness_ensemble': fit, 'acquisition': acq, 'ensemble_training': trn, 'kernels': kernels,
        }
    legs = None
    if world == 1 and not a.no_legs:         # N = 1 only, like cpu_baseline: the scaling runs time the headline alone
        del algo, pred, net
        torch.cuda.empty_cache()
        legs = config_legs(a, dist, device, backend, rank, world)
    if rank == 0:
        line['configs'] = legs
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
