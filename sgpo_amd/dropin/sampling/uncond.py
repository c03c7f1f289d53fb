"""Drop-in `sampling.uncond.Uncond` (reference: sampling/uncond.py:10-72) for the discrete
models: D3PM ancestral loop (uncond.py:56-62) and `MDLM.sample` (uncond.py:63-64).
Continuous / causal-LM nets are outside the hot-path scope."""
import torch
import tqdm

from models.d3pm import D3PM
from models.mdlm import MDLM

from .base import Algo


class Uncond(Algo):
    def __init__(self, net, forward_op, device='cuda', temp=1., top_p=1., data_config=None):
        super().__init__(net, forward_op, data_config=data_config, device=device)
        self.temp = temp
        self.top_p = top_p

    @torch.no_grad()
    def inference(self, seq_len=None, num_samples=1, verbose=True, detokenize=False, project=False):
        if type(self.net) == D3PM:
            timesteps = torch.linspace(self.net.timestep - 1, 1, int((self.net.timestep - 1) / 1), dtype=int)
            timesteps = tqdm.tqdm(timesteps) if verbose else timesteps
            x = self.net.get_start(num_samples)
            for t in timesteps:
                x = self.net.p_sample(x, t)
        elif type(self.net) == MDLM:
            x = self.net.sample(num_samples)
        else:
            raise ValueError(f'Model {type(self.net)} not supported for unconditional sampling on the HIP path')
        if detokenize:
            return x, self._untokenize_batch(x)
        return x
