"""ORACLE (test infrastructure, not product): CPU restatement of the classifier
guidance sampling loop.

Follows /root/reference/sampling/cls_guidance.py:26-69
(Classifier_Guidance_Inpaint.inference) and sampling/base.py:13-38 (mask /
full_seq / project).  `torch.multinomial(p, 1)` is restated as
`argmax(p / e)`, e ~ Exp(1) drawn elementwise (verified
bit-identical on CPU); `e` is injected.
Pinned by tests/golden/cg_*.npz.
"""
import torch

from . import mdlm as omdlm


def multinomial_from_exp(p, e):
    """torch.multinomial(p, 1) given the exponential draws e (same shape as p)."""
    return (p / e).argmax(dim=-1)


def cg_step(q, x, ratio_fn, e):
    """One guided step given the soft reverse probabilities q (cls_guidance.py:38-52).

    ratio_fn(q_minus) -> q_minus * prob_ratio (get_guided_rates); identity when temp == 0.
    """
    B, L, S = q.shape
    bi = torch.arange(B)[:, None]
    li = torch.arange(L)
    q = q.clone()
    q[bi, li, x] -= 1
    q = ratio_fn(q)
    q[bi, li, x] += 1
    q = q.clamp(min=0)
    p = q / q.sum(dim=-1, keepdim=True)
    return multinomial_from_exp(p.view(-1, S), e.view(-1, S)).view(B, L), p


def project(x, mask, full_seq):
    return x * mask + full_seq * (1 - mask)


def inference_mdlm(sd, B, L, T, mask_index, es, predictor=None, temperature=0.0, S=31,
                   emulate_bf16=False, return_trace=False, **model_kw):
    """MDLM + Classifier_Guidance_Inpaint.inference, T-1 steps; es[i] (B,L,S) Exp(1) draws.
    model_kw: diffusion / parameterization / time_conditioning of the net (configs/model/udlm.yaml for UDLM; the
    start state is all-mask for every diffusion, models/mdlm.py:52-53)."""
    timesteps = torch.linspace(T - 1, 1, int((T - 1) / 1), dtype=int)
    x = mask_index * torch.ones((B, L), dtype=torch.int64)
    trace = []
    for i, t in enumerate(timesteps):
        q = omdlm.p_sample(sd, x, t, T, mask_index, hard=False, emulate_bf16=emulate_bf16, **model_kw)

        def ratio_fn(qm):
            return omdlm.get_guided_rates(predictor, x, t, qm, S, guide_temp=temperature)[0]

        x, p = cg_step(q, x, ratio_fn, es[i])
        if return_trace:
            trace.append(x.clone())
    return (x, trace) if return_trace else x
