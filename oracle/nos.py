"""ORACLE (test infrastructure, not product): CPU restatement of the NOS hidden-state
optimisation step.

Follows /root/reference/models/mdlm.py:184-338 (MDLM._nos_denoise, absorbing state + subs
parameterisation) with the DITClassifier head on `x_emb` (dit.py:494-515: mean-pool over L,
Linear(hidden, 1)).  torch autograd + torch.optim.Adagrad as in the reference.
Pinned by tests/golden/nos_small.npz.
"""
import torch

from . import dit as odit
from . import mdlm as omdlm

NEG_INF = -1000000.0


def nos_denoise(sd, w_cls, b_cls, xt, t, T, mask_index, num_nos_steps, nos_step_size, nos_stability_coef, u,
                t_next_mode='reference', emulate_bf16=False):
    """emulate_bf16: the backbone pass rounds to bf16 where torch autocast does (the reference's GPU arithmetic, defect
    D7); the inner final-layer passes stay fp32.  Used to size the error a bf16 backbone alone puts into the result."""
    B = xt.shape[0]
    mt, ms, sig_t = omdlm.move_chances(t, T, B, t_next_mode)
    copy_flag = xt != mask_index
    sigma0 = torch.zeros_like(sig_t)                      # time_conditioning False (_process_sigma)
    with torch.no_grad():
        logits, hs = odit.dit_forward(sd, xt, sigma0, return_hidden_states=True, emulate_bf16=emulate_bf16)
        logits, h = logits.float(), hs[-1].float()
        log_x_theta = omdlm.subs_parameterization(logits, xt, mask_index)
        base = log_x_theta + torch.log(1. - (ms / mt))
        base[..., mask_index] = torch.log(ms / mt)[:, :, 0]
        base[copy_flag] = NEG_INF
        base[copy_flag, xt[copy_flag]] = 0.0
    kl_loss = torch.nn.KLDivLoss(log_target=True, reduction='batchmean')
    delta = torch.nn.Parameter(torch.zeros_like(h), requires_grad=True)
    opt = torch.optim.Adagrad([delta], lr=nos_step_size)
    with torch.enable_grad():
        for _ in range(num_nos_steps):
            h_cur = h + delta
            target = odit.dit_classifier_head(w_cls, b_cls, h_cur).sum()
            new_logits = omdlm.subs_parameterization(odit.dit_forward(sd, xt, sigma0, x_emb=h_cur), xt, mask_index)
            adj = new_logits + torch.log(1. - (ms / mt))
            adj[..., mask_index] = torch.log(ms / mt)[:, :, 0]
            adj[copy_flag] = NEG_INF
            adj[copy_flag, xt[copy_flag]] = 0.0
            loss = -target + nos_stability_coef * kl_loss(adj, base)
            opt.zero_grad()
            loss.backward()
            opt.step()
    with torch.no_grad():
        guided = omdlm.subs_parameterization(odit.dit_forward(sd, xt, sigma0, x_emb=h + delta.data), xt, mask_index)
        lp = guided + torch.log(1. - (ms / mt))
        lp[..., mask_index] = torch.log(ms / mt)[:, :, 0]
        probs = lp.exp()
        xs = omdlm.sample_categorical(probs, u)
        xs = torch.where(copy_flag, xt, xs)
    return xs, probs, delta.data
