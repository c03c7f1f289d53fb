"""Drop-in `models.mdlm.MDLM` (reference: models/mdlm.py:22-413) on the HIP library.

Same constructor and method signatures as the reference wrapper; every tensor
computation is a call into libsgpo_hip.so (sgpo_amd.runtime).  Defects of the
reference that shape results are reproduced by default:
  D1  `_sample_categorical` is the one of mdlm_diffusion.py:69-73
  D2  `t_next = t / T` re-divides the scaled t (models/mdlm.py:57-58); set
      `t_next_mode='corrected'` for the intended `(t-1)/T`
  D6  dropout is off (eval mode)
"""
import math
import os
from collections import deque

import torch

from sgpo_amd import _lib as L
from sgpo_amd import runtime as rt
from sgpo_amd import weights as W
from sgpo_amd.tokenizer import Tokenizer

from .base import GenerativeModel, cfg_get

EPS = 1e-3


class LogLinearNoise:
    """noise_schedule.py:126-144; returns (total_noise, rate_noise)."""

    def __init__(self, eps=EPS):
        self.eps = eps

    def __call__(self, t):
        t = torch.as_tensor(t, dtype=torch.float32)
        return -torch.log1p(-(1 - self.eps) * t), (1 - self.eps) / (1 - (1 - self.eps) * t)


class _Core:
    """Stands where the reference keeps its `MDLMDiffusion` (`net.model`): the attributes
    callers read (models/mdlm.py:36,53,59; pareto.py:95)."""

    def __init__(self, backbone, state_dict, config, tokenizer):
        self.backbone = backbone
        self._state_dict = state_dict
        self.config = config
        self.tokenizer = tokenizer
        self.vocab_size = len(tokenizer.alphabet)
        self.mask_index = tokenizer.mask_id
        self.noise = LogLinearNoise()
        self.neg_infinity = -1000000.0
        self.time_conditioning = bool(cfg_get(config, 'time_conditioning', False))
        self.parameterization = cfg_get(config, 'parameterization', 'subs')
        self.diffusion = cfg_get(config, 'diffusion', 'absorbing_state')
        self.length = cfg_get(config, 'model.length', 1024)

    def set_length(self, n):
        self.length = n

    def state_dict(self):
        return {'backbone.' + k: v for k, v in self._state_dict.items()}

    def eval(self):
        return self


def _instantiate_tokenizer(tokenizer):
    if isinstance(tokenizer, Tokenizer):
        return tokenizer
    return Tokenizer(sequences=True)  # the only tokenizer the configs name (evodiff.utils.Tokenizer)


def _load_dit_state(model_name, config):
    ckpt = os.path.join('checkpoints', str(model_name), 'best_model.ckpt')
    if os.path.exists(ckpt):
        return W.load_lightning_state(ckpt, 'backbone.')
    if str(model_name).startswith('synthetic') or os.environ.get('SGPO_SYNTHETIC_WEIGHTS') == '1':
        seed = int(os.environ.get('SGPO_SYNTHETIC_SEED', '0'))
        return W.dit_synthetic_state(
            seed=seed, vocab=31, hidden=int(cfg_get(config, 'model.hidden_size', 512)),
            cond_dim=int(cfg_get(config, 'model.cond_dim', 128)), n_blocks=int(cfg_get(config, 'model.n_blocks', 8)),
            n_heads=int(cfg_get(config, 'model.n_heads', 8)))
    raise FileNotFoundError(
        f'{ckpt} not found. Pretrained checkpoints are not shipped; pass model_name="synthetic..." or set '
        f'SGPO_SYNTHETIC_WEIGHTS=1 for seeded random weights of the configured architecture.')


class MDLM(GenerativeModel):
    NAN_WINDOW = 16

    def __init__(self, model_name, config, tokenizer, num_steps=128, seq_len=389, device='cuda', state_dict=None,
                 t_next_mode='reference', skip_unmasked=False):
        self.tokenizer = _instantiate_tokenizer(tokenizer)
        dev = rt.init(device if str(device) != 'cuda' else None)
        sd = state_dict if state_dict is not None else _load_dit_state(model_name, config)
        self.net = rt.DitDenoiser(sd, device=dev, max_len=max(int(cfg_get(config, 'model.length', 1024)), seq_len))
        self.model = _Core(self.net, sd, config, self.tokenizer)
        self.model.set_length(seq_len)
        self.S = self.model.vocab_size
        self.timestep = num_steps
        self.length = seq_len
        self.device = str(dev)
        self._dev = dev
        self.diffusion = cfg_get(config, 'diffusion', 'absorbing_state')
        self.parameterization = cfg_get(config, 'parameterization', 'subs')
        self.time_conditioning = bool(cfg_get(config, 'time_conditioning', False))
        self.t_next_mode = t_next_mode
        # Opt-in, output-identical: in the classifier-guidance step a sequence WITHOUT a masked position does not need the
        # denoiser (at an unmasked position the subs parameterisation replaces the logits by the carry-over one-hot,
        # mdlm_diffusion.py:338-340, so the step reads only the token, the schedule, the jump ratio and the noise).  With
        # defect D2 about (127/128)^L of the sequences are in that state at any step after the first (39 % at L = 119).
        # False (default) keeps the reference's one full forward per step.
        self.skip_unmasked = bool(skip_unmasked)
        self.skipped_rows = self.denoised_rows = 0
        self.global_batch = None        # set by a sharded caller: the unsharded batch size (NOS batchmean KL)
        self.noise_source = rt.NoiseSource(seed=torch.initial_seed())
        self.injected_noise = deque()   # parity runs: tensors consumed by successive sampling calls
        # one flag word per step of a check window: the sampling loop reads them every NAN_WINDOW steps and still names
        # the first offending step, as the reference raises AT that step (models/mdlm.py:410-411)
        self.NAN_WINDOW = max(type(self).NAN_WINDOW, int(num_steps))     # a recorded loop checks once, after its last step
        self._nan_flags = torch.zeros(self.NAN_WINDOW, dtype=torch.int32, device=dev)
        self._nan_flag = self._nan_flags[:1]
        # Opt-in: `Classifier_Guidance_Inpaint.inference` records its whole loop (T - 1 guided steps + projection) into ONE
        # HIP graph per (batch size, value function, temperature) and replays it with a new Philox key read from device
        # memory -- for small batches, where the ~70 launches of a step are launch-bound (GPU busy 0.78 at B = 64).
        # Same tokens as the eager loop, bit for bit.
        self.use_graphs = False

    # ------------------------------------------------------------------ noise bookkeeping
    def reseed(self, seed=None, row_offset=0):
        """New Philox key (default: torch.initial_seed(), i.e. what util/seed.py:set_seed set)."""
        self.noise_source = rt.NoiseSource(seed=torch.initial_seed() if seed is None else seed, row_offset=row_offset)

    def _next_noise(self):
        return self.injected_noise.popleft().to(self._dev, torch.float32) if self.injected_noise else None

    # ------------------------------------------------------------------ schedule
    def _move_chances(self, t, t_next=None):
        """(move_chance_t, move_chance_s, sigma_t) as fp32 host scalars (models/mdlm.py:57-68)."""
        tt = torch.as_tensor(t).detach().cpu().to(torch.float32) / self.timestep
        if self.t_next_mode == 'reference':
            tn = tt / self.timestep   # defect D2
        else:
            tn = (torch.as_tensor(t).detach().cpu().to(torch.float32) - 1) / self.timestep
        sig_t, _ = self.model.noise(tt)
        sig_s, _ = self.model.noise(tn)
        mt = 1 - torch.exp(-sig_t)
        ms = 1 - torch.exp(-sig_s)
        return float(mt), float(ms), sig_t

    def _denoise(self, x, sigma_t=None, out=None):
        """Raw denoiser logits (B,L,V); sigma only matters under time conditioning."""
        sig = None
        if self.time_conditioning and sigma_t is not None:
            sig = torch.as_tensor(sigma_t, dtype=torch.float32).reshape(-1)[:1].to(self._dev)
        return self.net.forward(x, sig, out=out)

    def _denoise_for_cg(self, x, sigma_t):
        """Logits for the classifier-guidance step.  skip_unmasked: the denoiser runs on the sequences that still hold a
        mask only (one device-to-host read of their count per step); the other rows of the returned buffer keep whatever
        they held -- the reverse-step kernel never uses them (reverse_step.hip: `lp` is overwritten where x != mask).  A
        row's logits do not depend on the batch it is in (tests: batch-size invariance), so the tokens are those of the
        full forward bit for bit."""
        if not (self.skip_unmasked and self.diffusion == 'absorbing_state'):
            return self._denoise(x, sigma_t)
        B = x.shape[0]
        idx = (x == self.model.mask_index).any(dim=1).nonzero().squeeze(1)
        n = int(idx.numel())
        self.denoised_rows += n
        self.skipped_rows += B - n
        buf = self.__dict__.get('_cg_logits')
        if buf is None or buf.shape[0] != B or buf.shape[1] != x.shape[1]:
            buf = self._cg_logits = torch.zeros(B, x.shape[1], self.S, dtype=torch.float32, device=self._dev)
        if n == B:
            return self._denoise(x, sigma_t, out=buf)
        if n > 0:
            buf.index_copy_(0, idx, self._denoise(x.index_select(0, idx), sigma_t))
        return buf

    # ------------------------------------------------------------------ reference API
    def sample(self, num_samples):
        """MDLMDiffusion._sample with the ddpm predictor + noise removal (mdlm_diffusion.py:661-700)."""
        x = self.get_start(num_samples)
        for i in range(self.timestep):
            x = self.ddpm_step(x, i)
        return self.noise_removal(x)

    def _sample_grid(self, eps=1e-5):
        """(timesteps, dt) of mdlm_diffusion.py:671-673: t runs from 1 down to eps in `self.timestep` updates."""
        return torch.linspace(1, eps, self.timestep + 1), (1 - eps) / self.timestep

    @torch.no_grad()
    def ddpm_step(self, x, i, want_probs=False):
        """Update i of the ancestral sampler, MDLMDiffusion._ddpm_update (mdlm_diffusion.py:634-658) at t = timesteps[i]
        with the correct t - dt schedule (not defect D2's): carry-over of unmasked tokens, categorical draw elsewhere."""
        timesteps, dt = self._sample_grid()
        sig_t, _ = self.model.noise(timesteps[i])
        sig_s, _ = self.model.noise(timesteps[i] - dt)
        mt, ms = float(1 - torch.exp(-sig_t)), float(1 - torch.exp(-sig_s))
        logits = self._denoise(x, sig_t)
        xo, q = rt.reverse_step_mdlm(logits, x, mt, ms, self.model.mask_index, L.STEP_HARD, noise_src=self.noise_source,
                                     noise=self._next_noise(), want_probs=want_probs)
        return (xo, q) if want_probs else xo

    @torch.no_grad()
    def noise_removal(self, x, want_probs=False):
        """The final argmax of the denoiser at t = eps (mdlm_diffusion.py:692-699)."""
        timesteps, _ = self._sample_grid()
        sig, _ = self.model.noise(timesteps[-1])
        logits = self._denoise(x, sig)
        xo, q = rt.reverse_step_mdlm(logits, x, 1.0, 0.0, self.model.mask_index, L.STEP_HARD, noise_mode=L.NOISE_NONE,
                                     want_probs=want_probs)
        return (xo, q) if want_probs else xo

    def get_start(self, batch_size):
        return torch.full((batch_size, self.length), self.model.mask_index, dtype=torch.int64, device=self._dev)

    @torch.no_grad()
    def p_sample(self, x, t, t_next=None, hard=True, guidance=None):
        mt, ms, sig_t = self._move_chances(t)
        x = x.to(self._dev)
        logits = self._denoise(x, sig_t)
        if self.diffusion == 'absorbing_state':
            if hard:
                xo, _ = rt.reverse_step_mdlm(logits, x, mt, ms, self.model.mask_index, L.STEP_HARD,
                                             noise_src=self.noise_source, noise=self._next_noise())
                return xo
            _, q = rt.reverse_step_mdlm(logits, x, mt, ms, self.model.mask_index, L.STEP_SOFT)
            return q
        if self.diffusion == 'uniform':
            if hard:
                xo, _ = rt.reverse_step_udlm(logits, x, mt, ms, L.STEP_HARD, noise_src=self.noise_source,
                                             noise=self._next_noise())
                return xo
            _, q = rt.reverse_step_udlm(logits, x, mt, ms, L.STEP_SOFT)
            return q
        raise NotImplementedError(f'Diffusion type {self.diffusion} not implemented.')

    def sigma(self, t):
        sigma, _ = self.model.noise(torch.as_tensor(t) / self.timestep)
        return sigma

    def q_sample(self, x, t):
        """Forward noising (models/mdlm.py:104-109); t (B,) integer timesteps."""
        sig, _ = self.model.noise(torch.as_tensor(t).to(torch.float32).cpu() / self.timestep)
        mc = 1 - torch.exp(-sig)
        if self.diffusion == 'absorbing_state':
            return rt.q_sample_mdlm(x.to(self._dev), mc, self.model.mask_index, noise_src=self.noise_source,
                                    noise=self._next_noise())
        if self.diffusion == 'uniform':     # mdlm_diffusion.py:598-601: torch.rand, then torch.randint(0, vocab)
            u = self._next_noise()
            tok = self.injected_noise.popleft() if u is not None else None
            return rt.q_sample_udlm(x.to(self._dev), mc, self.model.vocab_size, noise_src=self.noise_source, noise=u,
                                    noise_tokens=tok)
        raise NotImplementedError(f'Diffusion type {self.diffusion} not implemented.')

    # ------------------------------------------------------------------ guidance
    def _jump_log_ratios(self, predictor_model, x, t, skip_zero_rate=False):
        """(compact log-ratios (B,n_slots,S), slot_of_pos (L,) int32 or None).

        skip_zero_rate: only for the fused absorbing-state step, whose prior is exactly 0 wherever the
        skipped ratios would be used (include/sgpo_hip.h, sgpo_guidance_ratio_mlp_absorbing)."""
        vf = getattr(predictor_model, 'hip_value_function', None)
        if vf is not None:
            vf = vf()
            mi = self.model.mask_index if (skip_zero_rate and self.diffusion == 'absorbing_state') else None
            lr = vf.jump_log_ratios(x, float(t), self.S, mask_index=mi)
            return lr, _slot_map(vf.read_pos, x.shape[1], x.device)
        return _generic_jump_log_ratios(predictor_model, x, t, self.S), None

    def get_guided_rates(self, predictor_model, x, t, q_t, use_tag=False, guide_temp=1.0,
                         log_prob_ratio_cutoff=80.0):
        """Classifier-guided rates (models/mdlm.py:340-413): exact jump ratios, or their Taylor estimate with use_tag.

        The sampling loop does not call this: `Classifier_Guidance_Inpaint.inference` uses the
        fused kernel.  This entry point keeps the reference signature for other callers.
        """
        if guide_temp == 0:
            return q_t
        slots = None
        if use_tag:       # first-order estimate from one gradient (never enabled by the reference's callers)
            from ._tag import taylor_log_ratios
            lr = taylor_log_ratios(predictor_model, x, t, self.S)
        else:
            lr, slots = self._jump_log_ratios(predictor_model, x, t)
        if slots is not None:
            full = torch.zeros(x.shape[0], x.shape[1], self.S, dtype=torch.float32, device=x.device)
            pos = torch.nonzero(slots >= 0).squeeze(-1)
            full[:, pos] = lr[:, slots[pos].long()]
            lr = full
        q_t = q_t * torch.exp(torch.clamp(lr / guide_temp, max=80))
        if q_t.isnan().any():
            raise ValueError("The rate matrix 'q_t' contains NaNs.")
        return q_t

    @torch.no_grad()
    def guided_step(self, x, t, predictor_model=None, guide_temp=0.0, want_probs=False, check=True, step=None,
                    x_out=None, key_dev=None, key_offset=0, scratch=None):
        """One fused classifier-guidance step (sampling/cls_guidance.py:34-52): denoiser ->
        jump log-ratios -> reverse-step kernel.  Returns the next tokens (and p if asked).

        check: raise right here, as the reference does inside get_guided_rates / torch.multinomial, if the step met
        a NaN rate or a row it cannot sample from (one device-to-host read).  The sampling loop passes False and calls
        `check_nan()` every few steps instead; `step` (its loop index) selects the flag word, so that the deferred
        check still reports the step at which the rates went bad."""
        self._select_nan_slot(step)
        mt, ms, sig_t = self._move_chances(t)
        if key_dev is not None:
            # under HIP-graph capture (cls_guidance._GraphedLoop): static buffers from `scratch`, Philox key from device
            # memory, no host read
            assert self.diffusion == 'absorbing_state' and not check and not want_probs
            logits = self._denoise(x, sig_t, out=scratch['logits'])
            lr = slots = None
            if predictor_model is not None and guide_temp != 0:
                vf = predictor_model.hip_value_function()
                lr = vf.jump_log_ratios(x, float(t), self.S, out=scratch['lr'], mask_index=self.model.mask_index)
                slots = _slot_map(vf.read_pos, x.shape[1], x.device)
            xo, _ = rt.reverse_step_mdlm(logits, x, mt, ms, self.model.mask_index, L.STEP_SOFT_CG, log_ratio=lr,
                                         slot_of_pos=slots, guide_temp=guide_temp, nan_flag=self._nan_flag, x_out=x_out,
                                         key_dev=key_dev, key_offset=key_offset)
            return xo
        logits = self._denoise_for_cg(x, sig_t)
        lr = slots = None
        if predictor_model is not None and guide_temp != 0:
            lr, slots = self._jump_log_ratios(predictor_model, x, t, skip_zero_rate=True)
        xo, p = self._cg_reverse_step(logits, x, mt, ms, self.noise_source, self._next_noise(), lr, slots, guide_temp,
                                      want_probs=want_probs)
        if check:
            self.check_nan()
        return (xo, p) if want_probs else xo

    def _cg_reverse_step(self, logits, x, mt, ms, src, noise, lr, slots, guide_temp, want_probs=False, x_out=None):
        """The reverse-step kernel of the net's diffusion in classifier-guidance mode: absorbing state (models/mdlm.py:78-86)
        or the UDLM posterior (models/mdlm.py:87-95), then cls_guidance.py:38-52."""
        if self.diffusion == 'absorbing_state':
            return rt.reverse_step_mdlm(logits, x, mt, ms, self.model.mask_index, L.STEP_SOFT_CG, noise_src=src, noise=noise,
                                        log_ratio=lr, slot_of_pos=slots, guide_temp=guide_temp, want_probs=want_probs,
                                        nan_flag=self._nan_flag, x_out=x_out)
        if self.diffusion == 'uniform':
            return rt.guided_step_udlm(logits, x, mt, ms, noise_src=src, noise=noise, log_ratio=lr, slot_of_pos=slots,
                                       guide_temp=guide_temp, want_probs=want_probs, nan_flag=self._nan_flag, x_out=x_out)
        raise NotImplementedError(f'Diffusion type {self.diffusion} not implemented.')

    @torch.no_grad()
    def guided_step_groups(self, x, t, groups, guide_temp, step=None, runner=None, slots=None, x_out=None, keys_dev=None,
                           key_offset=0, logits_out=None):
        """One guided step for SEVERAL independent sampler calls stacked in one batch (Thompson draws of an ensemble,
        iterativeBO.py:204-237): ONE denoiser forward over all rows, then per group its own value function's jump ratios
        and its own reverse step with its own Philox stream.  groups: [(predictor_model, NoiseSource, lo, hi)], rows
        [lo, hi) of x.  A row's logits do not depend on the batch it is in (tests: batch-size invariance), so every group
        gets exactly the tokens its own `guided_step` call would have produced."""
        self._select_nan_slot(step)
        mt, ms, sig_t = self._move_chances(t)
        if keys_dev is not None:     # under HIP-graph capture (cls_guidance._GraphedManyLoop): static buffers, keys from device memory
            logits = self._denoise(x, sig_t, out=logits_out)
            return runner.step(logits, x, float(t), mt, ms, self.model.mask_index, guide_temp, slots, self._nan_flag, x_out,
                               keys_dev=keys_dev, key_offset=key_offset)
        logits = self._denoise_for_cg(x, sig_t)
        x_out = torch.empty_like(x)
        fused = self._fused_groups(groups, x.shape[1]) if self.diffusion == 'absorbing_state' else None
        if fused is not None:      # every group's value function lives in the library: one call steps them all
            runner, slots = fused
            return runner.step(logits.contiguous(), x.contiguous(), float(t), mt, ms, self.model.mask_index, guide_temp,
                               slots, self._nan_flag, x_out)
        for pred, src, lo, hi in groups:
            xs = x[lo:hi]
            lr = slots = None
            if pred is not None and guide_temp != 0:
                lr, slots = self._jump_log_ratios(pred, xs, t, skip_zero_rate=True)
            self._cg_reverse_step(logits[lo:hi], xs, mt, ms, src, None, lr, slots, guide_temp, x_out=x_out[lo:hi])
        return x_out

    def _fused_groups(self, groups, Lq):
        """(rt.GuidedStepGroups, slot map) for `groups` if each predictor offers `hip_value_function` (or is None) and
        the row ranges are consecutive from 0; None otherwise.  Cached on the groups list of a run."""
        cache = self.__dict__.setdefault('_groups_cache', {})
        key = id(groups)
        hit = cache.get(key)
        if hit is not None and hit[0] is groups:
            return hit[1]
        out = None
        vfs = []
        ok = groups[0][2] == 0 and all(groups[i][3] == groups[i + 1][2] for i in range(len(groups) - 1))
        for pred, _, _, _ in groups:
            vf = getattr(pred, 'hip_value_function', None) if pred is not None else None
            if pred is not None and vf is None:
                ok = False
                break
            vfs.append(vf() if vf is not None else None)
        live = [v for v in vfs if v is not None]
        if ok and all(v.read_pos == live[0].read_pos for v in live):
            runner = rt.GuidedStepGroups([(v, src, lo, hi) for v, (_, src, lo, hi) in zip(vfs, groups)], self.S, self._dev)
            slots = _slot_map(live[0].read_pos, Lq, self._dev) if live else None
            out = (runner, slots)
        cache.clear()              # one live run at a time; the entry keeps `groups` alive so its id stays unique
        cache[key] = (groups, out)
        return out

    def _select_nan_slot(self, step):
        slot = 0 if step is None else int(step) % self.NAN_WINDOW
        self._nan_flag = self._nan_flags[slot:slot + 1]
        if step is not None:
            self._nan_steps = getattr(self, '_nan_steps', {})
            self._nan_steps[slot] = int(step)

    def check_nan(self):
        """The reference's two failure modes of a guided step, read back from the kernels' flag words: NaN rates
        (`raise ValueError`, models/mdlm.py:410-411) and a row torch.multinomial would refuse (cls_guidance.py:52).
        Names the first offending step of the window (one flag word per step) and both conditions if both occurred."""
        flags = self._nan_flags.cpu().tolist()
        if not any(flags):
            return
        steps = getattr(self, '_nan_steps', {})
        bad = sorted((steps.get(slot, -1), slot) for slot, f in enumerate(flags) if f)
        step, slot = bad[0]
        flag = flags[slot]
        self._nan_flags.zero_()
        self._nan_steps = {}
        where = f' (first at reverse step index {step})' if step >= 0 else ''
        also = ' -- and the guided rates of a row sum to 0, inf or nan (torch.multinomial would refuse it)' if flag == 3 else ''
        if flag & 1:
            raise ValueError(f"The rate matrix 'q_t' contains NaNs.{where}{also}")
        raise RuntimeError(f'invalid multinomial distribution (sum of guided rates is 0, inf or nan){where}')

    def _nos_denoise(self, classifier_model, num_nos_steps, nos_step_size, nos_stability_coef, xt, t):
        from .nos_step import nos_denoise
        # KLDivLoss(reduction='batchmean') (models/mdlm.py:254,297) divides by the batch of the CALL: when a batch is
        # sharded over ranks the caller sets `net.global_batch` to the unsharded size
        return nos_denoise(self, classifier_model, num_nos_steps, nos_step_size, nos_stability_coef, xt, t,
                           kl_batch=getattr(self, 'global_batch', None))


_slot_cache = {}


def _slot_map(read_pos, Lq, device):
    key = (tuple(read_pos), Lq, str(device))
    if key not in _slot_cache:
        if list(read_pos) == list(range(Lq)):
            _slot_cache[key] = None
        else:
            m = torch.full((Lq,), -1, dtype=torch.int32)
            for i, p in enumerate(read_pos):
                m[p] = i
            _slot_cache[key] = m.to(device)
    return _slot_cache[key]


def _generic_jump_log_ratios(predictor_model, x, t, S, chunk=64):
    """Reference formulation for arbitrary predictors (e.g. GP samples): materialised single-site
    jumps, evaluated `chunk` sequences at a time (models/mdlm.py:111-162,360-375)."""
    B, D = x.shape
    out = torch.empty(B, D, S, dtype=torch.float32, device=x.device)
    jump_idx = torch.arange(D * S, device=x.device)
    dims, states = jump_idx // S, jump_idx % S
    for b0 in range(0, B, chunk):
        xb = x[b0:b0 + chunk]
        n = xb.shape[0]
        tt = t * torch.ones((n,), device=x.device)
        lp = predictor_model(xb, tt)
        xj = xb.unsqueeze(1).repeat(1, D * S, 1)
        xj[:, jump_idx, dims] = states
        lpj = predictor_model(xj.view(-1, D), tt.repeat_interleave(D * S)).view(n, D, S)
        out[b0:b0 + n] = lpj - lp.view(n, 1, 1)
    return out
