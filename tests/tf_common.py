"""Shared bookkeeping of the teacher-forced GPU parity tests (device step vs the CPU oracle at the same state with the
same injected noise).

Tolerances, as `north_star` states them -- denoiser/guided logits within bf16 rtol 1e-2, tokens bit-exact wherever the
race margin exceeds that tolerance:
  * The device returns probabilities p = softmax(logits) (x the fp32 guidance ratio), so a RELATIVE error of p is an
    ABSOLUTE error of a logit.  The logit tolerance is LOGIT_RTOL = 1e-2 of the largest reference logit of the step
    (max |z_ref|, about 3.3 at the real DiT dims, 2.0 at the small test dims), hence
        |p - p_ref| <= ATOL + (LOGIT_RTOL * max|z_ref|) * p_ref.
    ATOL = 2e-6 covers the fp32 cancellation of cls_guidance.py:38-49 ((q - 1) * ratio + 1 at the current token).
  * tokens: the categorical draw is argmax_k p[k] / e[k]; it is compared wherever the winner leads the runner-up by
    more than twice that relative tolerance (the winner may lose it, the runner-up may gain it), i.e. wherever an
    error of the stated size cannot flip the race.
The fraction of positions compared and the worst observed errors are printed (pytest -rP shows them)."""
import torch

DIM_KEYS = ('hidden', 'cond_dim', 'n_blocks', 'n_heads')
ALPHABET = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'
LOGIT_RTOL = 1e-2
ATOL = 2e-6


def oracle_soft_step(sd, xs, t, T, mask_index=28, diffusion='absorbing_state'):
    """The oracle's soft reverse probabilities q at state xs (== omdlm.p_sample(..., hard=False)) together with
    max |raw denoiser logit| of that state, the scale of the logit tolerance.  diffusion 'uniform' = the UDLM net of
    configs/model/udlm.yaml (d3pm parameterisation, time conditioned)."""
    from oracle import dit as odit
    from oracle import mdlm as omdlm
    if diffusion == 'uniform':
        _, _, sig_t = omdlm.move_chances(t, T, xs.shape[0])
        z = odit.dit_forward(sd, xs, sig_t)
        q = omdlm.p_sample(sd, xs, t, T, mask_index, diffusion='uniform', hard=False,
                           log_p_x0=omdlm.d3pm_parameterization(z, mask_index))
        return q, float(z.abs().max())
    z = odit.dit_forward(sd, xs, torch.zeros(xs.shape[0]))
    logp = omdlm.subs_parameterization(z, xs, mask_index)
    q = omdlm.p_sample(sd, xs, t, T, mask_index, hard=False, log_p_x0=logp)
    return q, float(z.abs().max())


class TeacherForced:
    def __init__(self, logit_rtol=LOGIT_RTOL, atol=ATOL, min_checked=0.8):
        self.logit_rtol, self.atol, self.min_checked = logit_rtol, atol, min_checked
        self.n_checked = self.n_total = 0
        self.worst = 0.0          # max |dp| / (atol + rtol_p p_ref): 1.0 is the tolerance
        self.worst_rel = 0.0      # max |dp| / p_ref over entries with p_ref > 1e-3
        self.max_scale = 0.0
        self.last_ok = None

    def step(self, i, p, p_ref, x, x_ref, e, logit_scale):
        rtol_p = self.logit_rtol * logit_scale
        self.max_scale = max(self.max_scale, logit_scale)
        d = (p - p_ref).abs()
        self.worst = max(self.worst, float((d / (self.atol + rtol_p * p_ref.abs())).max()))
        big = p_ref > 1e-3
        if bool(big.any()):
            self.worst_rel = max(self.worst_rel, float((d[big] / p_ref[big]).max()))
        torch.testing.assert_close(p, p_ref, rtol=rtol_p, atol=self.atol, msg=lambda m: f'step {i}: {m}')
        v = p_ref / e
        top2 = v.topk(2, dim=-1).values
        ok = (top2[..., 0] - top2[..., 1]) > 2.0 * rtol_p * top2[..., 0]
        self.n_checked += int(ok.sum())
        self.n_total += ok.numel()
        self.last_ok = ok
        assert torch.equal(x[ok], x_ref[ok]), f'step {i}: token mismatch outside the tie margin'
        return ok

    def finish(self, name):
        frac = self.n_checked / max(self.n_total, 1)
        print(f'[teacher-forced {name}] tokens compared bit-exact: {self.n_checked}/{self.n_total} = {frac:.4f}; '
              f'logit tolerance {self.logit_rtol:g} x max|z| (max|z| <= {self.max_scale:.2f}); worst |dp| / tolerance = '
              f'{self.worst:.3f}; worst relative error of p where p > 1e-3: {self.worst_rel:.2e}')
        assert frac > self.min_checked, frac
        return frac
