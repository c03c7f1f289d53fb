"""Taylor-approximated guidance (TAG), the `use_tag=True` branch of `get_guided_rates` (models/mdlm.py:377-394,
models/d3pm.py:145-...): the log-ratio of every single-site jump is estimated from ONE gradient of the value function at
the one-hot point instead of B*D*S evaluations.  Torch autograd through whatever predictor the caller hands in; it
must accept one-hot rows `(B, D, S)` float -- the drop-in `MLPModel` does (the reference's one-hots its input a second
time and cannot run this branch, SURVEY a-5)."""
import torch
import torch.nn.functional as F


def taylor_log_ratios(predictor_model, x, t, S):
    """(B, D, S): grad[b, d, s] - grad[b, d, x_bd], grad = d predictor(one-hot, t).sum() / d one-hot."""
    onehot = F.one_hot(x.long(), num_classes=S).to(torch.float32)
    tt = t * torch.ones((x.shape[0],), device=x.device)
    with torch.enable_grad():
        point = onehot.clone().requires_grad_(True)
        (grad,) = torch.autograd.grad(predictor_model(point, tt).sum(), point)
    return grad - (grad * onehot).sum(dim=-1, keepdim=True)
