"""Drop-in `models.d3pm.D3PM` (reference: models/d3pm.py:9-268) on the HIP library.

Denoiser = ByteNetLMTime (third party upstream, restated: parity unpinned); the transition
schedule is `Tokenizer.q_random_schedule` (restated).  The K x K
posterior of `p_sample` (models/d3pm.py:238-260, fp64, incl. defect D9) and `q_sample`
run as fused kernels; the 26 x 26 step matrices are chosen on the host exactly as the
reference does (fp32 torch ops on CPU, cached per (t, t_next)).
"""
import os
from collections import deque

import torch

from sgpo_amd import runtime as rt
from sgpo_amd import weights as W
from sgpo_amd.tokenizer import Tokenizer

from .base import GenerativeModel
from .mdlm import _generic_jump_log_ratios, _slot_map


class _Net:
    """`net.model` stand-in: callable (x, t) -> logits like ByteNetLMTime.forward."""

    def __init__(self, denoiser, state_dict):
        self._d = denoiser
        self._state_dict = state_dict

    def __call__(self, x, t):
        return self._d.forward(x, t)

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def state_dict(self):
        return self._state_dict


def _load_bytenet_state(model_name):
    ckpt = os.path.join('checkpoints', str(model_name), 'best_model.ckpt')
    if os.path.exists(ckpt):
        return W.load_lightning_state(ckpt, 'network.')
    if str(model_name).startswith('synthetic') or os.environ.get('SGPO_SYNTHETIC_WEIGHTS') == '1':
        return W.bytenet_synthetic_state(seed=int(os.environ.get('SGPO_SYNTHETIC_SEED', '0')))
    raise FileNotFoundError(
        f'{ckpt} not found. Pretrained checkpoints are not shipped (and evodiff.pretrained downloads are '
        f'unavailable offline); pass model_name="synthetic..." or set SGPO_SYNTHETIC_WEIGHTS=1.')


class D3PM(GenerativeModel):
    def __init__(self, model_name, seq_len, device='cuda', state_dict=None, schedule_accumulate='fp64'):
        dev = rt.init(device if str(device) != 'cuda' else None)
        self._dev = dev
        self.tokenizer = Tokenizer(sequences=True)
        self.S = self.tokenizer.K
        sd = state_dict if state_dict is not None else _load_bytenet_state(model_name)
        self.net = rt.BytenetDenoiser(sd, device=dev)
        self.model = _Net(self.net, sd)
        self.scheme = 'd3pm'
        self.timestep = 500
        Q_bar, Q = self.tokenizer.q_random_schedule(timesteps=self.timestep, accumulate=schedule_accumulate)
        self._Q_bar_cpu = Q_bar[:, :self.S, :self.S].contiguous()
        self._Q_cpu = Q[:, :self.S, :self.S].contiguous()
        self.Q_bar = self._Q_bar_cpu.to(dev)
        self.Q = self._Q_cpu.to(dev)
        self.device = str(dev)
        self.length = seq_len
        self._sigma_cpu = -torch.log(((self._Q_bar_cpu[:, 0, 0] - 1 / self.S) * self.S / (self.S - 1)))  # d3pm.py:29
        self.sigma = self._sigma_cpu.to(dev)
        self.noise_source = rt.NoiseSource(seed=torch.initial_seed())
        self.injected_noise = deque()
        self._step_cache = {}

    def reseed(self, seed=None, row_offset=0):
        self.noise_source = rt.NoiseSource(seed=torch.initial_seed() if seed is None else seed, row_offset=row_offset)

    def _next_noise(self):
        return self.injected_noise.popleft() if self.injected_noise else None

    def score(self, x, t):
        pass

    def pred_mean(self, x, t):
        """softmax of the first S logits (models/d3pm.py:34-37); compatibility entry, off the fused path."""
        return torch.softmax(self.net.forward(x, int(t))[:, :, :self.S], dim=-1)

    def get_start(self, batch_size):
        p = torch.full((batch_size, self.length, self.S), 1.0 / self.S, dtype=torch.float32, device=self._dev)
        nz = self._next_noise()
        if nz is not None and nz.dtype == torch.int64:   # parity runs inject the start tokens themselves
            return nz.to(self._dev)
        return rt.categorical_sample(p, noise_src=self.noise_source, noise=nz)

    def q_sample(self, x, t):
        return rt.q_sample_d3pm(x.to(self._dev), t, self.Q_bar, self.S, noise_src=self.noise_source,
                                noise=self._next_noise())

    def _step_matrices(self, t, t_next):
        """(Delta_Q, Q_bar') of models/d3pm.py:239-253 as device tensors."""
        key = (int(t), None if t_next is None else int(t_next))
        if key not in self._step_cache:
            t = key[0]
            if key[1] is None:
                dq = self._Q_cpu[t]
                qb = self._Q_bar_cpu[t - 1]
            else:
                ds = self._sigma_cpu[key[1]] - self._sigma_cpu[t]
                dq = torch.eye(self.S) * torch.exp(ds) + torch.ones(self.S) / self.S * (1 - torch.exp(ds))
                qb = self._Q_bar_cpu[key[1]]
            self._step_cache[key] = (dq.contiguous().to(self._dev), qb.contiguous().to(self._dev))
        return self._step_cache[key]

    @torch.no_grad()
    def p_sample(self, x, t, t_next=None, hard=True):
        x = x.to(self._dev)
        logits = self.net.forward(x, int(t))
        dq, qb = self._step_matrices(t, t_next)
        return rt.reverse_step_d3pm(logits, x, dq, qb, self.S, hard=hard, noise_src=self.noise_source,
                                    noise=self._next_noise() if hard else None)

    def get_guided_rates(self, predictor_model, x, t, q_t, use_tag=False, guide_temp=1.0,
                         log_prob_ratio_cutoff=80.0):
        """models/d3pm.py:109-199, exact branch; the jump log-ratios come from the fused kernel when the
        predictor is the MLP value function."""
        if guide_temp == 0:
            return q_t
        vf = None if use_tag else getattr(predictor_model, 'hip_value_function', None)
        if use_tag:       # first-order estimate from one gradient (never enabled by the reference's callers)
            from ._tag import taylor_log_ratios
            lr = taylor_log_ratios(predictor_model, x, t, self.S)
        elif vf is not None:
            vf = vf()
            lr_c = vf.jump_log_ratios(x, float(t), self.S)
            slots = _slot_map(vf.read_pos, x.shape[1], x.device)
            if slots is None:
                lr = lr_c
            else:
                lr = torch.zeros(x.shape[0], x.shape[1], self.S, dtype=torch.float32, device=x.device)
                pos = torch.nonzero(slots >= 0).squeeze(-1)
                lr[:, pos] = lr_c[:, slots[pos].long()]
        else:
            lr = _generic_jump_log_ratios(predictor_model, x, t, self.S)
        q_t = q_t * torch.exp(torch.clamp(lr / guide_temp, max=80))
        if q_t.isnan().any():
            raise ValueError("The rate matrix 'q_t' contains NaNs.")
        return q_t
