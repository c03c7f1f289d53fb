"""ORACLE (test infrastructure, not product): CPU fp32 restatement of the MDLM
wrapper and diffusion-core sampling functions.

Follows /root/reference:
  models/pretraining/model/mdlm/noise_schedule.py:126-144  LogLinearNoise
  models/pretraining/model/mdlm_diffusion.py:69-73         _sample_categorical
  models/pretraining/model/mdlm_diffusion.py:325-341       _subs_parameterization
  models/pretraining/model/mdlm_diffusion.py:343-348       _d3pm_parameterization
  models/pretraining/model/mdlm_diffusion.py:365-390       _process_sigma / forward
  models/pretraining/model/mdlm_diffusion.py:392-419       _compute_posterior (UDLM)
  models/pretraining/model/mdlm_diffusion.py:585-608       q_xt
  models/pretraining/model/mdlm_diffusion.py:634-700       _ddpm_update / _sample
  models/mdlm.py:55-109                                    p_sample / sigma / q_sample
  models/mdlm.py:111-162, 340-413                          jump transitions / guided rates

All randomness is injected (uniform tensors `u`) so the functions are
deterministic; goldens record the tensors the reference drew under the same seed.
Pinned by tests/golden/mdlm_*.npz.
"""
import torch
import torch.nn.functional as F

from . import dit as odit

NEG_INF = -1000000.0
EPS = 1e-3


def total_noise(t):
    """LogLinearNoise.total_noise (noise_schedule.py:143-144)."""
    return -torch.log1p(-(1 - EPS) * t)


def sample_categorical(probs, u):
    """_sample_categorical with the uniform draw injected (mdlm_diffusion.py:69-73)."""
    gumbel_norm = 1e-10 - (u + 1e-10).log()
    return (probs / gumbel_norm).argmax(dim=-1)


def subs_parameterization(logits, xt, mask_index):
    logits = logits.clone()
    logits[:, :, mask_index] += NEG_INF
    logits = logits - torch.logsumexp(logits, dim=-1, keepdim=True)
    unmasked = xt != mask_index
    logits[unmasked] = NEG_INF
    logits[unmasked, xt[unmasked]] = 0
    return logits


def d3pm_parameterization(logits, mask_index, subs_masking=False):
    logits = logits.clone()
    if subs_masking:
        logits[:, :, mask_index] += NEG_INF
    return logits - torch.logsumexp(logits, dim=-1, keepdim=True)


def forward(sd, x, sigma, mask_index, parameterization='subs', time_conditioning=False,
            x_emb=None, emulate_bf16=False):
    """MDLMDiffusion.forward: returns log p(x0) (mdlm_diffusion.py:376-390)."""
    if not time_conditioning:
        sigma = torch.zeros_like(sigma)
    logits = odit.dit_forward(sd, x, sigma, x_emb=x_emb, emulate_bf16=emulate_bf16)
    if parameterization == 'subs':
        return subs_parameterization(logits, x, mask_index)
    if parameterization == 'd3pm':
        return d3pm_parameterization(logits, mask_index)
    raise NotImplementedError(parameterization)


def compute_posterior(x, xt, alpha_s, alpha_t, vocab):
    """UDLM posterior (mdlm_diffusion.py:392-419)."""
    alpha_ts = alpha_t / alpha_s
    d_alpha = alpha_s - alpha_t
    oh = F.one_hot(xt, vocab)
    lim = torch.ones((1, 1, vocab), dtype=torch.float32) / vocab
    return ((alpha_t * vocab * x * oh + (alpha_ts - alpha_t) * oh + d_alpha * x
             + (1 - alpha_ts) * (1 - alpha_s) * lim)
            / (alpha_t * vocab * torch.gather(x, -1, xt[..., None]) + (1 - alpha_t)))


def move_chances(t, T, B, t_next_mode='reference'):
    """(move_chance_t, move_chance_s) of MDLM.p_sample (mdlm.py:57-70), incl. defect D2."""
    tt = torch.as_tensor(t) / T
    if t_next_mode == 'reference':
        tn = tt / T                       # mdlm.py:58 re-divides the already scaled t
    else:
        tn = (torch.as_tensor(t) - 1) / T
    sig_t = total_noise(tt) * torch.ones(B)
    sig_s = total_noise(tn) * torch.ones(B)
    mt = (1 - torch.exp(-sig_t))[:, None, None]
    ms = (1 - torch.exp(-sig_s))[:, None, None]
    return mt, ms, sig_t


def p_sample(sd, x, t, T, mask_index, diffusion='absorbing_state', parameterization='subs',
             time_conditioning=False, hard=True, u=None, t_next_mode='reference',
             emulate_bf16=False, log_p_x0=None):
    """MDLM.p_sample (mdlm.py:55-98).  `u` is the injected uniform draw for hard=True."""
    B = x.shape[0]
    mt, ms, sig_t = move_chances(t, T, B, t_next_mode)
    if log_p_x0 is None:
        log_p_x0 = forward(sd, x, sig_t, mask_index, parameterization, time_conditioning,
                           emulate_bf16=emulate_bf16)
    if diffusion == 'absorbing_state':
        q_xs = log_p_x0.exp() * (mt - ms)
        q_xs[:, :, mask_index] = ms[:, :, 0]
        if hard:
            _x = sample_categorical(q_xs, u)
            copy = (x != mask_index).to(x.dtype)
            return copy * x + (1 - copy) * _x
        return q_xs
    if diffusion == 'uniform':
        q_xs = compute_posterior(log_p_x0.exp(), x, 1 - ms, 1 - mt, log_p_x0.shape[-1])
        if hard:
            return sample_categorical(q_xs, u)
        return q_xs
    raise NotImplementedError(diffusion)


def q_sample(x, t, T, mask_index, u, diffusion='absorbing_state', uniform_tokens=None):
    """MDLM.q_sample -> q_xt (mdlm.py:104-109, mdlm_diffusion.py:585-601). t (B,) ints."""
    sigma = total_noise(t / T)
    move_chance = 1 - torch.exp(-sigma[:, None])
    move = u < move_chance
    if diffusion == 'absorbing_state':
        return torch.where(move, torch.as_tensor(mask_index), x)
    return torch.where(move, uniform_tokens, x)


def get_all_jump_transitions(x, S):
    """mdlm.py:111-162 / d3pm.py:55-106 (materialised B*D*S x D form)."""
    B, D = x.shape
    xe = x.unsqueeze(1).repeat(1, D * S, 1).view(-1, D)
    jump_idx = torch.arange(D * S).repeat(B, 1).flatten()
    xj = xe.clone()
    xj[torch.arange(jump_idx.size(0)), jump_idx // S] = jump_idx % S
    return xj


def get_guided_rates(predictor, x, t, q_t, S, guide_temp=1.0, use_tag=False):
    """get_guided_rates (mdlm.py:340-413, d3pm.py:109-199), exact branch and (use_tag) the first-order Taylor branch.

    exact: predictor(x (N,D), t (N,)) -> (N,1) on every materialised single-site jump.
    TAG (mdlm.py:377-394): predictor(one-hot (N,D,S) float, t) is differentiated at the one-hot point; the estimate of
    the jump (d, s) is grad[d, s] - grad[d, x_d].
    Returns (q_t * ratio, log_ratio_scaled_clamped).
    """
    if guide_temp == 0:
        return q_t, None
    B, D = x.shape
    tt = t * torch.ones((B,))
    if use_tag:
        onehot = torch.nn.functional.one_hot(x.long(), S).float()
        with torch.enable_grad():
            point = onehot.clone().requires_grad_(True)
            (grad,) = torch.autograd.grad(predictor(point, tt).sum(), point)
        lr = grad - (grad * onehot).sum(-1, keepdim=True)
    else:
        lp = predictor(x, tt)
        xj = get_all_jump_transitions(x, S)
        lpj = predictor(xj, tt.repeat(1, D * S).flatten()).view(B, D, S)
        lr = lpj - lp.view(B, 1, 1)
    lr = lr / guide_temp
    lr = torch.clamp(lr, max=80)
    q = q_t * torch.exp(lr)
    if q.isnan().any():
        raise ValueError("The rate matrix 'q_t' contains NaNs.")
    return q, lr


def ddpm_update(sd, x, t, dt, mask_index, u, time_conditioning=False):
    """_ddpm_update (mdlm_diffusion.py:634-658); t (B,1) in [0,1]."""
    sig_t = total_noise(t).squeeze(-1)
    sig_s = total_noise(t - dt).squeeze(-1)
    mt = (1 - torch.exp(-sig_t))[:, None, None]
    ms = (1 - torch.exp(-sig_s))[:, None, None]
    log_p = forward(sd, x, sig_t, mask_index, 'subs', time_conditioning)
    q_xs = log_p.exp() * (mt - ms)
    q_xs[:, :, mask_index] = ms[:, :, 0]
    _x = sample_categorical(q_xs, u)
    copy = (x != mask_index).to(x.dtype)
    return copy * x + (1 - copy) * _x


def sample(sd, B, L, num_steps, mask_index, us, eps=1e-5, time_conditioning=False):
    """_sample with predictor 'ddpm' + noise removal (mdlm_diffusion.py:661-700)."""
    x = mask_index * torch.ones(B, L, dtype=torch.int64)
    timesteps = torch.linspace(1, eps, num_steps + 1)
    dt = (1 - eps) / num_steps
    for i in range(num_steps):
        t = timesteps[i] * torch.ones(B, 1)
        x = ddpm_update(sd, x, t, dt, mask_index, us[i], time_conditioning)
    t = timesteps[-1] * torch.ones(B, 1)
    sig = total_noise(t).squeeze(-1)
    return forward(sd, x, sig, mask_index, 'subs', time_conditioning).argmax(dim=-1)
