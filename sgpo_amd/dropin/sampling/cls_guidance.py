"""Drop-in `sampling.cls_guidance.Classifier_Guidance_Inpaint`
(reference: sampling/cls_guidance.py:7-69).

With the HIP-backed MDLM the per-step chain  p_sample(soft) -> -1 at x ->
get_guided_rates -> +1 -> clamp -> normalise -> torch.multinomial  is ONE fused
kernel fed by the denoiser and the jump-ratio kernel (`net.guided_step`).  Any
other `net` (e.g. D3PM for the round-0 data of pareto.py:48-67) runs the same
algebra through `net.p_sample` / `net.get_guided_rates`.
"""
import torch
import tqdm

from .base import Algo


class Classifier_Guidance_Inpaint(Algo):
    NAN_CHECK_EVERY = 16

    def __init__(self, net, data_config, forward_op=None, temperature=1, n_max_mutations=None, device='cuda'):
        super().__init__(net=net, forward_op=forward_op, data_config=data_config, n_max_mutations=n_max_mutations,
                         device=device)
        self.temperature = temperature if forward_op is not None else 0.

    def update_model(self, classifier):
        self.forward_op = classifier.eval()  # make sure the model is in eval mode

    def project(self, x):
        return x * self.mask + self.full_seq * (1 - self.mask)

    @torch.no_grad()
    def inference(self, num_samples=1, verbose=False, detokenize=False, inpaint=False):
        timesteps = torch.linspace(self.net.timestep - 1, 1, int((self.net.timestep - 1) / 1), dtype=int)
        timesteps = tqdm.tqdm(timesteps) if verbose else timesteps
        x = self.net.get_start(num_samples)
        fused = hasattr(self.net, 'guided_step')
        check = getattr(self.net, 'check_nan', None) if fused else None
        for i, t in enumerate(timesteps):
            if fused:
                x = self.net.guided_step(x, t, self.forward_op, self.temperature, check=False, step=i)
                # the reference raises at the offending step (models/mdlm.py:410-411); reading the kernels' flags costs a
                # device sync, so they are read every NAN_CHECK_EVERY steps and after the last one -- one flag word per
                # step of the window, so the error still names the step
                if check is not None and (i + 1) % self.NAN_CHECK_EVERY == 0:
                    check()
            else:
                x = self._generic_step(x, t)
            if inpaint:
                x = self.project(x)
        if check is not None:
            check()
        x = self.project(x)
        if detokenize:
            detokenized = self.project_sequences(self._untokenize_batch(x))
            return x, detokenized
        return x

    @torch.no_grad()
    def inference_many(self, forward_ops, seeds, num_samples=1, inpaint=False):
        """`inference(num_samples)` for several value functions at once -- not in the reference, whose Thompson loop
        (iterativeBO.py:204-237) calls `update_model(member); inference(10)` once per draw: ten sequences are 1190 rows, a
        twentieth of what fills the GPU.  The draws are independent, so they are stacked: one denoiser forward per step
        for all of them, guidance and reverse step per draw with the draw's own Philox key `seeds[i]`.  Returns the list
        of token batches, each bit-identical to `net.reseed(seeds[i]); update_model(forward_ops[i]); inference(num_samples)`."""
        from sgpo_amd import runtime as rt
        n = len(forward_ops)
        assert n == len(seeds) and n > 0 and hasattr(self.net, 'guided_step_groups')
        timesteps = torch.linspace(self.net.timestep - 1, 1, int((self.net.timestep - 1) / 1), dtype=int)
        groups = [(op.eval() if op is not None else None, rt.NoiseSource(seed=sd, row_offset=0), i * num_samples, (i + 1) * num_samples)
                  for i, (op, sd) in enumerate(zip(forward_ops, seeds))]
        x = self.net.get_start(n * num_samples)
        for i, t in enumerate(timesteps):
            x = self.net.guided_step_groups(x, t, groups, self.temperature, step=i)
            if (i + 1) % self.NAN_CHECK_EVERY == 0:
                self.net.check_nan()
            if inpaint:
                x = self.project(x)
        self.net.check_nan()
        x = self.project(x)
        return [x[lo:hi] for _, _, lo, hi in groups]

    def _generic_step(self, x, t):
        """sampling/cls_guidance.py:34-52 for nets without a fused step."""
        from sgpo_amd import runtime as rt
        p = self.net.p_sample(x, t, hard=False)
        bi = torch.arange(p.shape[0], device=p.device)[:, None]
        li = torch.arange(p.shape[1], device=p.device)
        p[bi, li, x] -= 1
        p = self.net.get_guided_rates(self.forward_op, x, t, p, guide_temp=self.temperature)
        p[bi, li, x] += 1
        p = p.clamp(min=0)
        p = p / p.sum(dim=-1, keepdim=True)
        src = getattr(self.net, 'noise_source', None)
        nz = self.net._next_noise() if hasattr(self.net, '_next_noise') else None
        return rt.categorical_sample(p, noise_src=src, noise=nz)


# names of the reference module that are outside the sampling hot path -- Classifier_Guidance_Continuous (prior_sample.py:15, configs/algorithm/cls_guidance_continuous.yaml) --
# resolve to the reference's own definitions when its code base is on sys.path
from sgpo_amd.dropin._fallthrough import forward_missing_names  # noqa: E402

__getattr__ = forward_missing_names('sampling', 'cls_guidance')
