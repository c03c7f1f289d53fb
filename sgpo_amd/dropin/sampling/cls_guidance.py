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


class _GraphedLoop:
    """The whole loop of `Classifier_Guidance_Inpaint.inference` -- T - 1 guided steps and the projection -- recorded once
    into a HIP graph (torch stream capture; the library's hot calls allocate nothing and never synchronise) and replayed
    per call.  What changes between calls, the Philox key (seed, first global row, call-offset base), is read by the
    reverse-step kernel from a 4-word device buffer (`sgpo_reverse_step_mdlm_devkey`); the per-step scalars (move
    chances, call index) are kernel arguments recorded with the launch.  Host work per call: one 32-byte copy, one graph
    launch, one read of the NaN flags."""

    def __init__(self, algo, num_samples, inpaint):
        from sgpo_amd import runtime as rt
        net = algo.net
        dev = net._dev
        B, Lq = int(num_samples), int(net.length)
        self.algo, self.B = algo, B
        self.timesteps = torch.linspace(net.timestep - 1, 1, int((net.timestep - 1) / 1), dtype=int)
        self.key = torch.zeros(4, dtype=torch.int64, device=dev)
        self.key_host = torch.zeros(4, dtype=torch.int64).pin_memory()
        self.xa = torch.empty(B, Lq, dtype=torch.int64, device=dev)
        self.xb = torch.empty_like(self.xa)
        guided = algo.forward_op is not None and algo.temperature != 0
        vf = algo.forward_op.hip_value_function() if guided else None
        self.scratch = {'logits': torch.empty(B, Lq, net.S, dtype=torch.float32, device=dev),
                        'lr': torch.empty(B, vf.n_read, net.S, dtype=torch.float32, device=dev) if guided else None}
        self.inpaint = inpaint
        # warm-up off the capturing stream: workspaces, function attributes, the value function's device tables
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self.xa.fill_(net.model.mask_index)
            net.guided_step(self.xa, self.timesteps[0], algo.forward_op, algo.temperature, check=False, step=0, x_out=self.xb,
                            key_dev=self.key, key_offset=0, scratch=self.scratch)
            algo.project(self.xb)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        net._nan_flags.zero_()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._body()

    def _body(self):
        algo, net = self.algo, self.algo.net
        x = self.xa
        x.fill_(net.model.mask_index)                       # get_start (models/mdlm.py:52-53)
        for i, t in enumerate(self.timesteps):
            nxt = self.xb if x is self.xa else self.xa
            net.guided_step(x, t, algo.forward_op, algo.temperature, check=False, step=i, x_out=nxt, key_dev=self.key,
                            key_offset=i, scratch=self.scratch)
            x = nxt
            if self.inpaint:
                x.copy_(algo.project(x))
        return algo.project(x)

    def run(self):
        net = self.algo.net
        src = net.noise_source
        self.key_host.copy_(torch.tensor(src.key_words(net.length), dtype=torch.int64))
        self.key.copy_(self.key_host, non_blocking=True)
        src.calls += len(self.timesteps)                    # the eager loop draws one call offset per step
        self.graph.replay()
        net.check_nan()
        return self.out.clone()


class _GraphedManyLoop:
    """`inference_many` (stacked Thompson draws) as one recorded HIP graph.  The graph is recorded with value-function
    SLOTS of the library's own, one per guided draw; a call copies the draws' ensemble members into the slots
    (`sgpo_mlp_copy`, device to device) and their Philox keys into the key buffer, then replays."""

    def __init__(self, algo, forward_ops, num_samples, inpaint):
        from sgpo_amd import runtime as rt
        from models.mdlm import _slot_map
        net = algo.net
        dev = net._dev
        n, b, Lq = len(forward_ops), int(num_samples), int(net.length)
        self.algo, self.n, self.b, self.inpaint = algo, n, b, inpaint
        self.timesteps = torch.linspace(net.timestep - 1, 1, int((net.timestep - 1) / 1), dtype=int)
        guided = algo.temperature != 0
        self.slots_vf = []
        for op in forward_ops:
            if op is None or not guided:
                self.slots_vf.append(None)
                continue
            m = op.model
            sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
            self.slots_vf.append(rt.MlpValueFunction(sd, m.read_positions(), alphabet=m.alphabet_size, hidden=m.hidden_dim,
                                                     time_conditioned=m.time_conditioned, device=dev))
        live = [v for v in self.slots_vf if v is not None]
        self.runner = rt.GuidedStepGroups([(v, None, i * b, (i + 1) * b) for i, v in enumerate(self.slots_vf)], net.S, dev)
        self.slot_map = _slot_map(live[0].read_pos, Lq, dev) if live else None
        self.keys = torch.zeros(4 * n, dtype=torch.int64, device=dev)
        self.keys_host = torch.zeros(4 * n, dtype=torch.int64).pin_memory()
        self.xa = torch.empty(n * b, Lq, dtype=torch.int64, device=dev)
        self.xb = torch.empty_like(self.xa)
        self.logits = torch.empty(n * b, Lq, net.S, dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self.xa.fill_(net.model.mask_index)
            self._step(self.xa, self.xb, 0, self.timesteps[0])
            algo.project(self.xb)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        net._nan_flags.zero_()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            x = self.xa
            x.fill_(net.model.mask_index)
            for i, t in enumerate(self.timesteps):
                nxt = self.xb if x is self.xa else self.xa
                self._step(x, nxt, i, t)
                x = nxt
                if inpaint:
                    x.copy_(algo.project(x))
            self.out = algo.project(x)

    def _step(self, x, nxt, i, t):
        algo = self.algo
        algo.net.guided_step_groups(x, t, None, algo.temperature, step=i, runner=self.runner, slots=self.slot_map, x_out=nxt,
                                    keys_dev=self.keys, key_offset=i, logits_out=self.logits)

    def run(self, forward_ops, seeds):
        net = self.algo.net
        words = []
        for i, (op, sd) in enumerate(zip(forward_ops, seeds)):
            if self.slots_vf[i] is not None:
                self.slots_vf[i].copy_from(op.eval().hip_value_function())
            v = int(sd) & 0xFFFFFFFFFFFFFFFF
            words += [v - (1 << 64) if v >= (1 << 63) else v, 0, 0, 0]      # NoiseSource(seed=sd, row_offset=0), first call
        self.keys_host.copy_(torch.tensor(words, dtype=torch.int64))
        self.keys.copy_(self.keys_host, non_blocking=True)
        self.graph.replay()
        net.check_nan()
        out = self.out.clone()
        return [out[i * self.b:(i + 1) * self.b] for i in range(self.n)]


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
        if self._graph_eligible(verbose):
            x = self._graphed(num_samples, inpaint).run()
            if detokenize:
                return x, self.project_sequences(self._untokenize_batch(x))
            return x
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

    def _graph_eligible(self, verbose):
        """The recorded loop serves the fused absorbing-state step with in-kernel Philox and a library value function."""
        net = self.net
        if verbose or not getattr(net, 'use_graphs', False) or getattr(net, 'diffusion', None) != 'absorbing_state':
            return False
        if getattr(net, 'injected_noise', None) or getattr(net, 'skip_unmasked', False) or getattr(net, 'time_conditioning', False):
            return False
        guided = self.forward_op is not None and self.temperature != 0
        return not guided or hasattr(self.forward_op, 'hip_value_function')

    def _graphed(self, num_samples, inpaint):
        """The recorded loop for this (batch, value function and its weights, temperature); recorded on first use, and
        again when any of them changes (one live graph per sampler object)."""
        guided = self.forward_op is not None and self.temperature != 0
        key = (int(num_samples), bool(inpaint), float(self.temperature), self.net.t_next_mode,
               id(self.forward_op) if guided else None, self.forward_op._weights_key() if guided else None)
        hit = self.__dict__.get('_graph_cache')
        if hit is None or hit[0] != key:
            self._graph_cache = None                       # free the old graph's buffers before recording the new one
            self._graph_cache = hit = (key, _GraphedLoop(self, num_samples, inpaint))
        return hit[1]

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
        if self._graph_eligible(False) and all(op is None or hasattr(op, 'hip_value_function') for op in forward_ops):
            # one recorded loop per (draw count, batch, which draws are guided, architecture of the value functions)
            sig = tuple(None if (op is None or self.temperature == 0) else
                        (tuple(op.model.read_positions()), op.model.alphabet_size, op.model.hidden_dim, op.model.time_conditioned)
                        for op in forward_ops)
            live = [s_ for s_ in sig if s_ is not None]
            if not live or all(s_ == live[0] for s_ in live):
                key = ('many', n, int(num_samples), bool(inpaint), float(self.temperature), self.net.t_next_mode, sig)
                hit = self.__dict__.get('_graph_cache_many')
                if hit is None or hit[0] != key:
                    self._graph_cache_many = None
                    self._graph_cache_many = hit = (key, _GraphedManyLoop(self, forward_ops, num_samples, inpaint))
                return hit[1].run(forward_ops, seeds)
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

    def _generic_step(self, x, t, want_probs=False):
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
        xo = rt.categorical_sample(p, noise_src=src, noise=nz)
        return (xo, p) if want_probs else xo


# names of the reference module that are outside the sampling hot path -- Classifier_Guidance_Continuous (prior_sample.py:15, configs/algorithm/cls_guidance_continuous.yaml) --
# resolve to the reference's own definitions when its code base is on sys.path
from sgpo_amd.dropin._fallthrough import forward_missing_names  # noqa: E402

__getattr__ = forward_missing_names('sampling', 'cls_guidance')
