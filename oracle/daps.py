"""ORACLE (test infrastructure, not product): CPU restatement of the discrete DAPS loop.

Follows /root/reference/sampling/daps.py:
  log_ratio           :68-73
  metropolis_hasting  :75-114  (noise injected: per step idx draw, value draw, uniform)
  uncond_sample       :116-124
  inference           :127-164
Pinned by tests/golden/daps_*.npz (captured from the reference import).
"""
import torch

from . import d3pm as od3pm


def log_ratio(sigma, hm_dist, S, length):
    alpha = (1 - torch.exp(-sigma)) * (1 - 1 / S)
    log_alpha = torch.log(alpha + 1e-5)
    log_1alpha = torch.log(1 - alpha)
    return hm_dist * log_alpha + (length - hm_dist) * log_1alpha


def metropolis_hasting(x0hat, log_likelihood, sigma, steps, alpha, beta, residues, S, length, r_idx, r_val, r_u):
    """r_idx (steps,N) ints in [0,n_res), r_val (steps,N) ints in [0,S), r_u (steps,N) uniforms."""
    residues = torch.tensor(residues)
    x = x0hat.clone()
    cur_ll = log_likelihood(x) * alpha
    cur_hm = (x != x0hat).sum(dim=-1)
    for s in range(steps):
        proposal = x.clone()
        idx = residues[r_idx[s]]
        proposal.scatter_(1, idx[:, None], r_val[s].unsqueeze(1))
        ll = log_likelihood(proposal) * alpha
        hm = (proposal != x0hat).sum(dim=-1)
        lr = ll - cur_ll
        lr = lr + (log_ratio(sigma / beta, hm, S, length) - log_ratio(sigma / beta, cur_hm, S, length))
        rho = torch.clip(torch.exp(lr), max=1.0)
        seed = r_u[s]
        x = x * (seed > rho).unsqueeze(-1) + proposal * (seed < rho).unsqueeze(-1)
        cur_ll = ll * (seed < rho) + cur_ll * (seed > rho)
        cur_hm = hm * (seed < rho) + cur_hm * (seed > rho)
    return x


def time_steps(T, num_steps):
    return torch.linspace(T - 1, 1, num_steps, dtype=int)


def uncond_sample(denoise, xt, t_start, ode_steps, Q, Q_bar, sigma, S, es):
    """es: list of (N*L,S) f64 Exp(1) draws, one per inner step."""
    ts = torch.linspace(int(t_start), 0, ode_steps, dtype=int)
    for k, (t, t_next) in enumerate(zip(ts[:-1], ts[1:])):
        logits = denoise(xt, int(t))
        xt = od3pm.p_sample(logits, xt, int(t), int(t_next), Q, Q_bar, sigma, S, e=es[k])
    return xt
