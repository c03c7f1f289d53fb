"""`MDLM._nos_denoise` (reference: models/mdlm.py:184-338) on the fused NOS kernels.

Implements the intended math (defect D5: the shipped NOS config cannot run): absorbing-state
diffusion + subs parameterisation; the classifier contributes through its `output_layer`
(Linear(hidden, 1) after mean-pooling `x_emb`, dit.py:494-515), so its gradient w.r.t. the
hidden state is the constant `weight / L`.
"""
import torch

from sgpo_amd import runtime as rt


def _head_weight(classifier_model, hidden):
    layer = getattr(classifier_model, 'output_layer', None)
    if layer is None and hasattr(classifier_model, 'model'):
        layer = getattr(classifier_model.model, 'output_layer', None)
    w = getattr(layer, 'weight', None)
    if w is None or w.numel() != hidden:
        raise TypeError('NOS needs a classifier with an `output_layer` = Linear(hidden, 1) applied to the mean-pooled '
                        f'hidden state (DITClassifier, dit.py:464); got {type(classifier_model).__name__}')
    return w


def nos_denoise(mdlm, classifier_model, num_nos_steps, nos_step_size, nos_stability_coef, xt, t, kl_batch=None):
    if mdlm.diffusion != 'absorbing_state' or mdlm.parameterization != 'subs':
        raise NotImplementedError('NOS on the HIP path covers absorbing-state diffusion with subs parameterisation')
    mt, ms, sig_t = mdlm._move_chances(t)
    sigma = None
    if mdlm.time_conditioning:
        sigma = torch.as_tensor(sig_t, dtype=torch.float32).reshape(-1)[:1]
    w = _head_weight(classifier_model, mdlm.net.hidden)
    xs, probs, _ = rt.nos_denoise(mdlm.net, xt.to(mdlm._dev), w, num_nos_steps, nos_step_size, nos_stability_coef, mt, ms,
                                  mdlm.model.mask_index, kl_batch=kl_batch, sigma=sigma, noise_src=mdlm.noise_source,
                                  noise=mdlm._next_noise())
    return xs, probs, None
