"""ORACLE (test infrastructure, not product): CPU restatement of the D3PM wrapper.

Follows /root/reference/models/d3pm.py:
  __init__ (Q, Q_bar slicing, sigma)   :16-29
  pred_mean                            :34-37
  q_sample                             :45-52
  p_sample (K x K posterior, fp64)     :230-268  (incl. defect D9: p0 enters squared)
`evodiff.utils.Tokenizer.q_random_schedule` is third party and absent: restated
from the published EvoDiff algorithm -- that part is
"parity unpinned".  The wrapper arithmetic itself is pinned by
tests/golden/d3pm_*.npz (captured from the reference import with a stand-in
denoiser and this schedule).
"""
import torch
import torch.nn.functional as F


def q_random_schedule(K=26, timesteps=500, accumulate='fp64'):
    dt = torch.float64 if accumulate == 'fp64' else torch.float32
    eye = torch.eye(K, dtype=dt)
    prod = eye.clone()
    qs, qbars = [], []
    for t in range(1, timesteps + 1):
        beta = 1.0 / (timesteps - t + 1)
        q = torch.full((K, K), beta / K, dtype=dt) + eye * (1.0 - beta)
        prod = prod @ q
        qs.append(q.float())
        qbars.append(prod.float())
    return torch.stack(qbars), torch.stack(qs)


def sigma_from_qbar(Q_bar, S):
    """d3pm.py:29."""
    return -torch.log(((Q_bar[:, 0, 0] - 1 / S) * S / (S - 1)))


def posterior(p0, x, t, t_next, Q, Q_bar, sigma, S):
    """p_theta_marg of D3PM.p_sample given p0 = pred_mean(x,t).to(float64) (d3pm.py:239-260)."""
    N, L = x.shape
    if t_next is None:
        Delta_Q = Q[t]
    else:
        ds = sigma[t_next] - sigma[t]
        Delta_Q = torch.eye(S) * torch.exp(ds) + torch.ones(S) / S * (1 - torch.exp(ds))
    A = Delta_Q.T[x]
    Qb = Q_bar[t - 1] if t_next is None else Q_bar[t_next]
    Q_expand = Qb.unsqueeze(0).unsqueeze(0).expand(N, L, S, S)
    B_pred = p0.unsqueeze(3) * Q_expand
    q_t = A.unsqueeze(2) * B_pred
    p = torch.matmul(q_t.transpose(2, 3), p0.unsqueeze(3)).squeeze(3)
    return p / p.sum(dim=2, keepdim=True)


def p_sample(logits, x, t, t_next, Q, Q_bar, sigma, S, e=None, hard=True):
    """D3PM.p_sample with the denoiser logits given; e = Exp(1) draws (N*L,S) f64."""
    p0 = F.softmax(logits[:, :, :S], dim=-1).to(torch.float64)
    p = posterior(p0, x, t, t_next, Q, Q_bar, sigma, S)
    if not hard:
        return p
    N, L = x.shape
    return (p.reshape(-1, S) / e.reshape(-1, S)).argmax(dim=-1).view(N, L)


def q_sample(x, t, Q_bar, S, e):
    """D3PM.q_sample (d3pm.py:45-52): t (B,) int64; e Exp(1) draws (B*L,S) f32."""
    oh = F.one_hot(x, num_classes=S).float()
    prob = torch.bmm(Q_bar[t].float(), oh.permute(0, 2, 1)).permute(0, 2, 1)
    return (prob.reshape(-1, S) / e.reshape(-1, S)).argmax(dim=-1).view(prob.shape[:-1])
