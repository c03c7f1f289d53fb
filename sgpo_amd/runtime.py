"""Thin torch-side owners of the library handles and workspaces.

PyTorch is plumbing here (device memory, the current HIP stream); every
computation below is a call into libsgpo_hip.so.
"""
import ctypes as C
import itertools

import torch

from . import _lib as L
from . import weights as W

_initialised = set()


def init(device=None):
    """Bind this process to a GPU (one rank per GPU) and load the library."""
    lib = L.load()
    if not torch.cuda.is_available():
        raise L.SgpoError('no GPU visible: the SGPO hot path runs on MI355X only (no CPU fallback)')
    dev = torch.device(device if device is not None else 'cuda')
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _initialised:
        torch.cuda.set_device(idx)
        L.check(lib.sgpo_init(idx), 'sgpo_init')
        _initialised.add(idx)
    return torch.device('cuda', idx)


DIT_ORDER_HEAD = ['vocab_embed.embedding', 'sigma_map.mlp.0.weight', 'sigma_map.mlp.0.bias',
                  'sigma_map.mlp.2.weight', 'sigma_map.mlp.2.bias']
DIT_ORDER_BLOCK = ['norm1.weight', 'attn_qkv.weight', 'attn_out.weight', 'norm2.weight', 'mlp.0.weight',
                   'mlp.0.bias', 'mlp.2.weight', 'mlp.2.bias', 'adaLN_modulation.weight', 'adaLN_modulation.bias']
DIT_ORDER_TAIL = ['output_layer.norm_final.weight', 'output_layer.linear.weight', 'output_layer.linear.bias',
                  'output_layer.adaLN_modulation.weight', 'output_layer.adaLN_modulation.bias']
MLP_ORDER = ['time_embedder.0.W', 'time_embedder.1.weight', 'time_embedder.1.bias', 'embedder.weight',
             'embedder.bias', 'mlp.0.weight', 'mlp.0.bias', 'mlp.2.weight', 'mlp.2.bias', 'mlp.4.weight',
             'mlp.4.bias', 'cls_head.0.weight', 'cls_head.0.bias', 'cls_head.2.weight', 'cls_head.2.bias']


def _blob(sd, keys, device):
    return torch.cat([sd[k].detach().to(torch.float32).reshape(-1) for k in keys]).to(device).contiguous()


class Workspace:
    """Grow-only device scratch buffer owned by the caller side (torch allocator)."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, nbytes):
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=self.device)
        return self.buf


class DitDenoiser:
    """DIT.forward on the GPU (dit.py:375-418) from a reference-keyed state dict."""

    def __init__(self, state_dict, device=None, max_len=1024):
        self.device = init(device)
        self.lib = L.load()
        d = W.dit_dims(state_dict)
        self.dims = d
        self.vocab, self.hidden = d['vocab'], d['hidden']
        self.cfg = L.DitConfig(d['vocab'], d['hidden'], d['cond_dim'], d['freq_dim'], d['n_blocks'], d['n_heads'],
                               d['mlp_hidden'], max_len)
        blob = _blob(state_dict, dit_keys(d['n_blocks']), self.device)
        n = self.lib.sgpo_dit_param_count(C.byref(self.cfg))
        if blob.numel() != n:
            raise L.SgpoError(f'DiT state dict has {blob.numel()} parameters, library expects {n}')
        h = C.c_void_p()
        L.check(self.lib.sgpo_dit_create(C.byref(self.cfg), L.ptr(blob), blob.numel(), C.byref(h)), 'sgpo_dit_create')
        self.handle = h
        self.ws = Workspace(self.device)

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.sgpo_dit_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def forward(self, tokens, sigma=None, x_emb=None, return_hidden=False, out=None):
        """tokens (B,L) int64; sigma None (=0), (1,) or (B,) fp32 -> logits (B,L,V) fp32."""
        B, Lq = tokens.shape
        tokens = tokens.contiguous()
        assert tokens.dtype == torch.int64 and tokens.is_cuda
        need = self.lib.sgpo_dit_workspace_bytes(self.handle, B, Lq)
        ws = self.ws.get(need)
        base = (ws.data_ptr() + 255) & ~255
        logits = out if out is not None else torch.empty(B, Lq, self.vocab, dtype=torch.float32, device=self.device)
        hidden = torch.empty(B, Lq, self.hidden, dtype=torch.float32, device=self.device) if return_hidden else None
        if sigma is not None:
            sigma = sigma.to(self.device, torch.float32).reshape(-1).contiguous()
        if x_emb is not None:
            x_emb = x_emb.to(self.device, torch.float32).contiguous()
        L.check(self.lib.sgpo_dit_forward(self.handle, L.ptr(tokens), L.ptr(sigma),
                                          0 if sigma is None else sigma.numel(), B, Lq, L.ptr(x_emb), L.ptr(logits),
                                          L.ptr(hidden), C.c_void_p(base), need, L.stream_ptr()), 'sgpo_dit_forward')
        return (logits, hidden) if return_hidden else logits


def nos_denoise(net, xt, w_cls, num_nos_steps, nos_step_size, nos_stability_coef, mt, ms, mask_index, kl_batch=None,
                sigma=None, noise_src=None, noise=None, want_probs=True, want_delta=False, hidden_in=None):
    """MDLM._nos_denoise on the GPU (net: DitDenoiser).  Returns (xs, guided_probs | None, delta | None)."""
    lib = L.load()
    B, Lq = xt.shape
    xt = xt.contiguous()
    dev = net.device
    w_cls = w_cls.detach().to(dev, torch.float32).reshape(-1).contiguous()
    assert w_cls.numel() == net.hidden
    need_d = lib.sgpo_dit_workspace_bytes(net.handle, B, Lq)
    need_n = lib.sgpo_nos_workspace_bytes(net.handle, B, Lq)
    if not hasattr(net, 'ws_nos'):
        net.ws_nos = Workspace(dev)
    wd, wn = net.ws.get(need_d), net.ws_nos.get(need_n)
    bd, bn = (wd.data_ptr() + 255) & ~255, (wn.data_ptr() + 255) & ~255
    if sigma is not None:
        sigma = sigma.to(dev, torch.float32).reshape(-1).contiguous()
    if hidden_in is not None:
        hidden_in = hidden_in.to(dev, torch.float32).contiguous()
    if noise is not None:
        noise = noise.to(dev, torch.float32).contiguous()
        noise_mode = L.NOISE_UNIFORM
    else:
        noise_mode = L.NOISE_PHILOX
    seed, off, row_off = (noise_src.seed, noise_src.next_offset(), noise_src.row_offset) if noise_src else (0, 0, 0)
    xs = torch.empty_like(xt)
    probs = torch.empty(B, Lq, net.vocab, dtype=torch.float32, device=dev) if want_probs else None
    delta = torch.empty(B, Lq, net.hidden, dtype=torch.float32, device=dev) if want_delta else None
    L.check(lib.sgpo_nos_denoise(net.handle, L.ptr(xt), L.ptr(sigma), 0 if sigma is None else sigma.numel(), B, Lq,
                                 L.ptr(w_cls), int(num_nos_steps), float(nos_step_size), float(nos_stability_coef),
                                 float(mt), float(ms), int(kl_batch or B), mask_index, noise_mode, L.ptr(noise), seed,
                                 off, row_off * Lq, L.ptr(hidden_in), L.ptr(xs), L.ptr(probs), L.ptr(delta),
                                 C.c_void_p(bn), need_n,
                                 C.c_void_p(bd), need_d, L.stream_ptr()), 'sgpo_nos_denoise')
    return xs, probs, delta


class MlpValueFunction:
    """MLPModel (problem/protein.py:58-121) on the GPU, incl. the factored jump-ratio kernel."""

    def __init__(self, state_dict, read_pos, alphabet=31, hidden=64, time_conditioned=True, device=None):
        self.device = init(device)
        self.lib = L.load()
        self.read_pos = [int(p) for p in read_pos]
        self.n_read = len(self.read_pos)
        self.alphabet = alphabet
        blob = _blob(state_dict, MLP_ORDER, self.device)
        arr = (C.c_int * self.n_read)(*self.read_pos)
        h = C.c_void_p()
        L.check(self.lib.sgpo_mlp_create(L.ptr(blob), blob.numel(), self.n_read, arr, alphabet, hidden,
                                         1 if time_conditioned else 0, C.byref(h)), 'sgpo_mlp_create')
        self.handle = h
        self.ws = Workspace(self.device)

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.sgpo_mlp_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def copy_from(self, other):
        """Overwrite this value function's device tables with `other`'s (same shape), asynchronously on the current stream."""
        L.check(self.lib.sgpo_mlp_copy(self.handle, other.handle, L.stream_ptr()), 'sgpo_mlp_copy')

    def forward(self, x, t=None):
        B, Lq = x.shape
        x = x.contiguous()
        if max(self.read_pos) >= Lq:
            raise L.SgpoError(f'value function reads position {max(self.read_pos)} of a length-{Lq} sequence')
        out = torch.empty(B, dtype=torch.float32, device=self.device)
        if t is not None:
            t = t.to(self.device, torch.float32).reshape(-1).contiguous()
            assert t.numel() == B
        L.check(self.lib.sgpo_mlp_forward(self.handle, L.ptr(x), L.ptr(t), B, Lq, L.ptr(out), L.stream_ptr()),
                'sgpo_mlp_forward')
        return out

    def jump_log_ratios(self, x, t, S, out=None, mask_index=None):
        """log_ratio (B, n_read, S) = f(x with x[read_pos[i]] <- s, t) - f(x, t).

        mask_index (absorbing-state diffusion only): skip the jumps whose prior rate is exactly 0, i.e. at
        unmasked positions evaluate only the jump to the mask and leave the other entries 0."""
        B, Lq = x.shape
        x = x.contiguous()
        if max(self.read_pos) >= Lq:
            raise L.SgpoError(f'value function reads position {max(self.read_pos)} of a length-{Lq} sequence')
        need = self.lib.sgpo_mlp_workspace_bytes(self.handle, B)
        ws = self.ws.get(need)
        base = (ws.data_ptr() + 255) & ~255
        if out is None:
            out = torch.empty(B, self.n_read, S, dtype=torch.float32, device=self.device)
        if mask_index is None:
            L.check(self.lib.sgpo_guidance_ratio_mlp(self.handle, L.ptr(x), float(t), B, Lq, S, L.ptr(out),
                                                     C.c_void_p(base), need, L.stream_ptr()), 'sgpo_guidance_ratio_mlp')
        else:
            L.check(self.lib.sgpo_guidance_ratio_mlp_absorbing(self.handle, L.ptr(x), float(t), B, Lq, S, int(mask_index),
                                                               L.ptr(out), C.c_void_p(base), need, L.stream_ptr()),
                    'sgpo_guidance_ratio_mlp_absorbing')
        return out


class MlpEnsembleTrainer:
    """Adam/MSE training of an ensemble of MLPModel value functions in lockstep on the GPU
    (training/train_classifier.py:7-64).  states: list of MLPModel state dicts."""

    def __init__(self, states, read_pos, alphabet=31, hidden=64, time_conditioned=True, device=None):
        self.device = init(device)
        self.lib = L.load()
        self.read_pos = [int(p) for p in read_pos]
        self.n_members = len(states)
        self.time_conditioned = bool(time_conditioned)
        self._shapes = [(k, tuple(states[0][k].shape)) for k in MLP_ORDER]
        blob = torch.stack([_blob(sd, MLP_ORDER, self.device) for sd in states]).contiguous()
        self.n_params = blob.shape[1]
        want = self.lib.sgpo_mlp_trainer_param_count(len(self.read_pos), alphabet, 1 if time_conditioned else 0)
        if self.n_params != want:
            raise L.SgpoError(f'MLPModel state dict has {self.n_params} parameters, the trainer expects {want}')
        arr = (C.c_int * len(self.read_pos))(*self.read_pos)
        h = C.c_void_p()
        L.check(self.lib.sgpo_mlp_trainer_create(L.ptr(blob), self.n_members, len(self.read_pos), arr, alphabet, hidden,
                                                 1 if time_conditioned else 0, C.byref(h)), 'sgpo_mlp_trainer_create')
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.sgpo_mlp_trainer_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def step(self, x, t, y, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        """One optimiser step for every member.  x (E, B, L) int64, t (E, B) or None, y (E, B); returns the
        members' MSE losses (E,) on the device (before the update)."""
        E, B, Lq = x.shape
        assert E == self.n_members and tuple(y.shape) == (E, B)
        if max(self.read_pos) >= Lq:
            raise L.SgpoError(f'value function reads position {max(self.read_pos)} of a length-{Lq} sequence')
        x = x.to(self.device, torch.int64).contiguous()
        y = y.to(self.device, torch.float32).contiguous()
        tt = None
        if self.time_conditioned:
            tt = t.to(self.device, torch.float32).contiguous()
            assert tuple(tt.shape) == (E, B)
        loss = torch.empty(E, dtype=torch.float32, device=self.device)
        L.check(self.lib.sgpo_mlp_trainer_step(self.handle, L.ptr(x), L.ptr(tt), L.ptr(y), B, Lq, float(lr), float(betas[0]),
                                               float(betas[1]), float(eps), L.ptr(loss), L.stream_ptr()), 'sgpo_mlp_trainer_step')
        return loss

    def steps(self, x, t, y, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        """n consecutive optimiser steps in one library call: x (n, E, B, L) int64, t (n, E, B) or None, y (n, E, B);
        returns the losses (n, E).  Same kernels and bits as n calls of `step`."""
        n, E, B, Lq = x.shape
        assert E == self.n_members and tuple(y.shape) == (n, E, B)
        if max(self.read_pos) >= Lq:
            raise L.SgpoError(f'value function reads position {max(self.read_pos)} of a length-{Lq} sequence')
        x = x.to(self.device, torch.int64).contiguous()
        y = y.to(self.device, torch.float32).contiguous()
        tt = None
        if self.time_conditioned:
            tt = t.to(self.device, torch.float32).contiguous()
            assert tuple(tt.shape) == (n, E, B)
        loss = torch.empty(n, E, dtype=torch.float32, device=self.device)
        L.check(self.lib.sgpo_mlp_trainer_steps(self.handle, L.ptr(x), L.ptr(tt), L.ptr(y), n, B, Lq, float(lr), float(betas[0]),
                                                float(betas[1]), float(eps), L.ptr(loss), L.stream_ptr()), 'sgpo_mlp_trainer_steps')
        return loss

    def state_dicts(self):
        blob = torch.empty(self.n_members, self.n_params, dtype=torch.float32, device=self.device)
        L.check(self.lib.sgpo_mlp_trainer_read(self.handle, L.ptr(blob), L.stream_ptr()), 'sgpo_mlp_trainer_read')
        out = []
        for e in range(self.n_members):
            sd, o = {}, 0
            for k, shp in self._shapes:
                n = 1
                for q in shp:
                    n *= q
                sd[k] = blob[e, o:o + n].reshape(shp).clone()
                o += n
            out.append(sd)
        return out


class FitnessEnsemble:
    """The fitness-oracle ensemble of oracle/inference_oracle.py on the GPU.

    states: list of OracleModel state dicts (fc1.weight, fc1.bias, fc2.weight, fc2.bias)."""

    def __init__(self, states, device=None):
        self.device = init(device)
        self.lib = L.load()
        self.n_members = len(states)
        self.hidden, in_dim = states[0]['fc1.weight'].shape
        assert in_dim % 20 == 0
        self.n_res = in_dim // 20
        parts = []
        for sd in states:
            assert tuple(sd['fc1.weight'].shape) == (self.hidden, in_dim)
            parts += [sd['fc1.weight'].reshape(-1), sd['fc1.bias'].reshape(-1), sd['fc2.weight'].reshape(-1),
                      sd['fc2.bias'].reshape(-1)]
        blob = torch.cat([q.to(torch.float32).cpu() for q in parts]).to(self.device).contiguous()
        h = C.c_void_p()
        L.check(self.lib.sgpo_fitness_create(L.ptr(blob), blob.numel(), self.n_members, self.n_res, self.hidden,
                                             C.byref(h)), 'sgpo_fitness_create')
        self.handle = h
        self.ws = Workspace(self.device)
        self._bad = torch.zeros(1, dtype=torch.int32, device=self.device)

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.sgpo_fitness_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def forward(self, tokens, parent_tokens=None, cutoff=0, rate=1.0):
        """tokens (N, n_res) int64 indices into 'ACDEFGHIKLMNPQRSTVWY'; parent_tokens (n_res,) or None
        (impose_penalty=False).  Returns float64 (N,).  Raises ValueError for a token outside the
        alphabet, like `alphabet.index` in the reference (oracle/train_oracle.py:45)."""
        N, n_res = tokens.shape
        if n_res != self.n_res:
            raise L.SgpoError(f'fitness ensemble trained on {self.n_res} residues, got {n_res}')
        tokens = tokens.to(self.device, torch.int64).contiguous()
        par = None if parent_tokens is None else parent_tokens.to(self.device, torch.int64).contiguous()
        out = torch.empty(N, dtype=torch.float64, device=self.device)
        need = self.lib.sgpo_fitness_workspace_bytes(self.handle, N)
        ws = self.ws.get(need)
        base = (ws.data_ptr() + 255) & ~255
        L.check(self.lib.sgpo_fitness_forward(self.handle, L.ptr(tokens), N, L.ptr(par), int(cutoff), float(rate),
                                              L.ptr(out), L.ptr(self._bad), C.c_void_p(base), need, L.stream_ptr()),
                'sgpo_fitness_forward')
        if int(self._bad.item()) != 0:
            self._bad.zero_()
            raise ValueError('substring not found')     # str.index's message in the reference
        return out


def project_tokens(tokens, full_seq=None, n_std=20, n_special=11, u_special=None, n_max_mutations=None, keys=None):
    """Token-level `Algo.project_sequences` (sampling/base.py:40-63), in place on
    tokens (N, L) int64.  u_special (N, n_special) uniforms replace special tokens; keys (N, L) iid uniforms
    pick which mutations survive the n_max_mutations cut.  Raises ValueError on a token outside the alphabet
    or a special token with no u_special."""
    dev = init(tokens.device if tokens.is_cuda else None)
    lib = L.load()
    assert tokens.dtype == torch.int64 and tokens.is_cuda and tokens.is_contiguous() and tokens.dim() == 2
    N, Ls = tokens.shape
    fs = None if full_seq is None else full_seq.to(dev, torch.int64).contiguous()
    us = None if u_special is None else u_special.to(dev, torch.float32).contiguous()
    ks = None if keys is None else keys.to(dev, torch.float32).contiguous()
    if us is not None:
        assert tuple(us.shape) == (N, n_special)
    nm = -1 if n_max_mutations is None else int(n_max_mutations)
    if nm >= 0:
        assert fs is not None and ks is not None and tuple(ks.shape) == (N, Ls) and fs.numel() == Ls
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    L.check(lib.sgpo_project_tokens(L.ptr(tokens), L.ptr(fs), N, Ls, n_std, n_special, L.ptr(us), nm, L.ptr(ks),
                                    L.ptr(bad), L.stream_ptr()), 'sgpo_project_tokens')
    b = int(bad.item())
    if b & 1:
        raise ValueError('token outside the alphabet')
    if b & 2:
        raise ValueError('special token in the batch and no u_special to replace it')
    return tokens


class AcquisitionSet:
    """The sequence store of `ProteinPredictorDataset` (dataset/protein.py:41-44,153-183) as a hash set in HBM
    (SURVEY.md section 8 row f-2).  `update` mirrors update_data's filter on a token batch."""

    KEEP_ALL, NOT_IN_STORE, NOT_IN_ROUND = 0, 1, 2

    def __init__(self, seq_len, capacity, max_batch, device=None):
        self.device = init(device)
        self.lib = L.load()
        self.seq_len, self.capacity, self.max_batch = int(seq_len), int(capacity), int(max_batch)
        h = C.c_void_p()
        L.check(self.lib.sgpo_acq_create(self.seq_len, self.capacity, self.max_batch, C.byref(h)), 'sgpo_acq_create')
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.sgpo_acq_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def __len__(self):
        return int(self.lib.sgpo_acq_count(self.handle))

    def update(self, tokens, n_needed, scope, round=0):
        """tokens (N, L) int64 on the device.  Returns (kept_idx (n_kept,) int64 batch rows appended, in batch
        order; n_admissible = rows that passed the filter before the n_needed cut)."""
        N = int(tokens.shape[0])
        if N == 0:
            return torch.empty(0, dtype=torch.int64, device=self.device), 0
        assert tokens.dtype == torch.int64 and tokens.is_cuda and tokens.shape[1] == self.seq_len
        tokens = tokens.contiguous()
        kept = torch.empty(N, dtype=torch.int64, device=self.device)
        nk, na = C.c_int(0), C.c_int(0)
        L.check(self.lib.sgpo_acq_update(self.handle, L.ptr(tokens), N, int(n_needed), int(scope), int(round),
                                         L.ptr(kept), C.byref(nk), C.byref(na), L.stream_ptr()), 'sgpo_acq_update')
        return kept[:nk.value], na.value

    def read(self, first=0, count=None):
        """(tokens (count, L) int64, rounds (count,) int32) of store rows [first, first+count)."""
        n = len(self)
        count = n - first if count is None else count
        tok = torch.empty(count, self.seq_len, dtype=torch.int64, device=self.device)
        rnd = torch.empty(count, dtype=torch.int32, device=self.device)
        L.check(self.lib.sgpo_acq_read(self.handle, first, count, L.ptr(tok), L.ptr(rnd), L.stream_ptr()), 'sgpo_acq_read')
        return tok, rnd


def landscape_lookup(table, tokens, residues, n_std=20):
    """update_data's 4-site branch (dataset/protein.py:188-197): fitness of each row's combo from a dense
    table (n_std ** n_res,) fp32; raises ValueError for a token outside the n_std letters."""
    dev = init(tokens.device if tokens.is_cuda else None)
    lib = L.load()
    tokens = tokens.to(dev, torch.int64).contiguous()
    table = table.to(dev, torch.float32).contiguous()
    res_host = [int(r) for r in residues]
    N, Ls = tokens.shape
    if min(res_host) < 0 or max(res_host) >= Ls:
        raise L.SgpoError(f'landscape_lookup: residue positions {res_host} outside a length-{Ls} row')
    res = torch.tensor(res_host, dtype=torch.int32).to(dev).contiguous()
    out = torch.empty(N, dtype=torch.float32, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    L.check(lib.sgpo_landscape_lookup(L.ptr(table), table.numel(), L.ptr(tokens), N, Ls, L.ptr(res), res.numel(), n_std,
                                      L.ptr(out), L.ptr(bad), L.stream_ptr()), 'sgpo_landscape_lookup')
    if int(bad.item()):
        raise ValueError('token outside the standard alphabet')
    return out


class NoiseSource:
    """Counter-based noise bookkeeping: one (seed, call offset) pair per sampling kernel call.

    Philox counters are keyed by the GLOBAL row index (row_offset = first row of this rank's
    shard), so a batch sharded over ranks draws the same numbers as the unsharded batch.
    """

    def __init__(self, seed=42, row_offset=0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.row_offset = int(row_offset)
        self.calls = 0

    def next_offset(self):
        self.calls += 1
        return self.calls - 1

    def key_words(self, seq_len):
        """(seed, first global row, call-offset base, 0): the 4-word device key of the *_devkey entry points for a loop
        that starts now.  Seeds >= 2^63 are stored by their two's-complement bit pattern (int64 tensors)."""
        sgn = lambda v: v - (1 << 64) if v >= (1 << 63) else v
        return [sgn(self.seed), self.row_offset * int(seq_len), self.calls, 0]


def reverse_step_mdlm(logits, x, mt, ms, mask_index, step_mode, noise_src=None, noise=None, noise_mode=None,
                      log_ratio=None, slot_of_pos=None, guide_temp=0.0, want_probs=False, nan_flag=None, x_out=None,
                      L_seq=None, key_dev=None, key_offset=0):
    lib = L.load()
    B, Lq, K = logits.shape
    x = x.contiguous()
    logits = logits.contiguous()
    if key_dev is not None:      # Philox key read from device memory (HIP-graph replay): call index `key_offset` baked in
        assert noise is None and key_dev.dtype == torch.int64 and key_dev.numel() >= 4 and key_dev.is_cuda
        probs = torch.empty_like(logits) if want_probs else None
        if x_out is None:
            x_out = torch.empty_like(x)
        n_slots = Lq if log_ratio is None else log_ratio.shape[1]
        L.check(lib.sgpo_reverse_step_mdlm_devkey(L.ptr(logits), L.ptr(x), L.ptr(log_ratio), L.ptr(slot_of_pos), n_slots, B, Lq,
                                                  K, mask_index, float(mt), float(ms), float(guide_temp), step_mode,
                                                  int(key_offset), L.ptr(key_dev), L.ptr(x_out), L.ptr(probs), L.ptr(nan_flag),
                                                  L.stream_ptr()), 'sgpo_reverse_step_mdlm_devkey')
        return x_out, probs
    if noise is not None:
        assert noise.dtype == torch.float32
        noise = noise.to(logits.device).contiguous()    # a host pointer handed to a kernel is a GPU fault
        if noise_mode is None:
            noise_mode = L.NOISE_EXPONENTIAL if step_mode == L.STEP_SOFT_CG else L.NOISE_UNIFORM
    elif noise_mode is None:
        noise_mode = L.NOISE_PHILOX
    seed, off, row_off = (noise_src.seed, noise_src.next_offset(), noise_src.row_offset) if noise_src else (0, 0, 0)
    probs = torch.empty_like(logits) if (want_probs or step_mode == L.STEP_SOFT) else None
    if step_mode != L.STEP_SOFT and x_out is None:
        x_out = torch.empty_like(x)
    n_slots = Lq if log_ratio is None else log_ratio.shape[1]
    L.check(lib.sgpo_reverse_step_mdlm(L.ptr(logits), L.ptr(x), L.ptr(log_ratio), L.ptr(slot_of_pos), n_slots, B, Lq, K,
                                       mask_index, float(mt), float(ms), float(guide_temp), step_mode, noise_mode,
                                       L.ptr(noise), seed, off, row_off * Lq, L.ptr(x_out), L.ptr(probs),
                                       L.ptr(nan_flag), L.stream_ptr()), 'sgpo_reverse_step_mdlm')
    return x_out, probs


class GuidedStepGroups:
    """Stacked sampler calls (Thompson draws) stepped through `sgpo_guided_step_groups_mdlm`: the per-group host arrays
    are built once, a step only refreshes the Philox call offsets.  groups: [(MlpValueFunction | None, NoiseSource, lo, hi)]
    with consecutive row ranges."""

    def __init__(self, groups, S, device):
        self.lib = L.load()
        self.groups = groups
        n = len(groups)
        assert n > 0 and all(groups[i][3] == groups[i + 1][2] for i in range(n - 1)), 'row ranges must be consecutive'
        vfs = [g[0] for g in groups if g[0] is not None]
        self.read_pos = vfs[0].read_pos if vfs else None
        assert all(v.read_pos == self.read_pos for v in vfs), 'stacked value functions must read the same positions'
        self.members = (C.c_void_p * n)(*[getattr(g[0].handle, 'value', g[0].handle) if g[0] is not None else None for g in groups])
        self.row_lo = (C.c_int * (n + 1))(*([g[2] for g in groups] + [groups[-1][3]]))
        self.seeds = (C.c_uint64 * n)(*[g[1].seed if g[1] is not None else 0 for g in groups])
        self.offsets = (C.c_uint64 * n)()
        self.n = n
        self.S = S
        rows = groups[-1][3]            # the scratch buffers cover the whole stacked batch
        self.ws_bytes = max([int(self.lib.sgpo_mlp_workspace_bytes(v.handle, rows)) for v in vfs] or [0]) + 256
        self.lr_bytes = rows * (len(self.read_pos) if vfs else 0) * S * 4
        self.ws = torch.empty(self.ws_bytes + 256, dtype=torch.uint8, device=device)
        self.lr = torch.empty(max(self.lr_bytes, 4), dtype=torch.uint8, device=device)

    def step(self, logits, x, t, mt, ms, mask_index, guide_temp, slot_of_pos, nan_flag, x_out, keys_dev=None, key_offset=0):
        B, Lq, K = logits.shape
        assert logits.is_contiguous() and x.is_contiguous() and x_out.is_contiguous()
        assert self.row_lo[0] == 0 and self.row_lo[self.n] == B
        if keys_dev is not None:     # the draws' Philox keys live in device memory (HIP-graph replay)
            assert keys_dev.dtype == torch.int64 and keys_dev.numel() >= 4 * self.n and keys_dev.is_cuda
            base = (self.ws.data_ptr() + 255) & ~255
            L.check(self.lib.sgpo_guided_step_groups_mdlm_devkey(
                self.n, self.members, self.row_lo, L.ptr(logits), L.ptr(x), Lq, K, self.S, float(t), int(mask_index), float(mt),
                float(ms), float(guide_temp), int(key_offset), L.ptr(keys_dev), L.ptr(slot_of_pos), L.ptr(self.lr), self.lr_bytes,
                C.c_void_p(base), self.ws_bytes, L.ptr(x_out), L.ptr(nan_flag), L.stream_ptr()),
                'sgpo_guided_step_groups_mdlm_devkey')
            return x_out
        row_offsets = (C.c_uint64 * self.n)(*[g[1].row_offset * Lq for g in self.groups])
        for i, g in enumerate(self.groups):
            self.offsets[i] = g[1].next_offset()
        base = (self.ws.data_ptr() + 255) & ~255
        L.check(self.lib.sgpo_guided_step_groups_mdlm(
            self.n, self.members, self.row_lo, L.ptr(logits), L.ptr(x), Lq, K, self.S, float(t), int(mask_index), float(mt),
            float(ms), float(guide_temp), self.seeds, self.offsets, row_offsets, L.ptr(slot_of_pos), L.ptr(self.lr),
            self.lr_bytes, C.c_void_p(base), self.ws_bytes, L.ptr(x_out), L.ptr(nan_flag), L.stream_ptr()),
            'sgpo_guided_step_groups_mdlm')
        return x_out


def reverse_step_udlm(logits, x, mt, ms, step_mode, noise_src=None, noise=None, want_probs=False):
    lib = L.load()
    B, Lq, K = logits.shape
    x = x.contiguous()
    logits = logits.contiguous()
    noise_mode = L.NOISE_PHILOX if noise is None else L.NOISE_UNIFORM
    if noise is not None:
        noise = noise.to(logits.device, torch.float32).contiguous()
    seed, off, row_off = (noise_src.seed, noise_src.next_offset(), noise_src.row_offset) if noise_src else (0, 0, 0)
    probs = torch.empty_like(logits) if (want_probs or step_mode == L.STEP_SOFT) else None
    x_out = torch.empty_like(x) if step_mode != L.STEP_SOFT else None
    L.check(lib.sgpo_reverse_step_udlm(L.ptr(logits), L.ptr(x), B, Lq, K, float(mt), float(ms), step_mode, noise_mode,
                                       L.ptr(noise), seed, off, row_off * Lq, L.ptr(x_out), L.ptr(probs),
                                       L.stream_ptr()), 'sgpo_reverse_step_udlm')
    return x_out, probs


def guided_step_udlm(logits, x, mt, ms, noise_src=None, noise=None, log_ratio=None, slot_of_pos=None, guide_temp=0.0,
                     want_probs=False, nan_flag=None, x_out=None):
    """sampling/cls_guidance.py:34-52 on a uniform-diffusion net: posterior -> CG reweight -> multinomial, one launch.
    noise: injected Exp(1) draws (B, L, K) for parity runs."""
    lib = L.load()
    B, Lq, K = logits.shape
    x = x.contiguous()
    logits = logits.contiguous()
    noise_mode = L.NOISE_PHILOX if noise is None else L.NOISE_EXPONENTIAL
    if noise is not None:
        assert noise.dtype == torch.float32
        noise = noise.to(logits.device).contiguous()
    seed, off, row_off = (noise_src.seed, noise_src.next_offset(), noise_src.row_offset) if noise_src else (0, 0, 0)
    probs = torch.empty_like(logits) if want_probs else None
    if x_out is None:
        x_out = torch.empty_like(x)
    if log_ratio is not None:
        log_ratio = log_ratio.contiguous()
        assert log_ratio.dtype == torch.float32 and log_ratio.shape[0] == B and log_ratio.shape[2] == K
    n_slots = Lq if log_ratio is None else log_ratio.shape[1]
    L.check(lib.sgpo_guided_step_udlm(L.ptr(logits), L.ptr(x), L.ptr(log_ratio), L.ptr(slot_of_pos), n_slots, B, Lq, K,
                                      float(mt), float(ms), float(guide_temp), noise_mode, L.ptr(noise), seed, off,
                                      row_off * Lq, L.ptr(x_out), L.ptr(probs), L.ptr(nan_flag), L.stream_ptr()),
            'sgpo_guided_step_udlm')
    return x_out, probs


def q_sample_udlm(x, move_chance, vocab, noise_src=None, noise=None, noise_tokens=None):
    """q_xt's uniform branch (mdlm_diffusion.py:593-601); noise (B, L) uniforms + noise_tokens (B, L) int64 for parity runs."""
    lib = L.load()
    B, Lq = x.shape
    x = x.contiguous()
    move_chance = move_chance.to(x.device, torch.float32).reshape(-1).contiguous()
    assert move_chance.numel() == B
    noise_mode = L.NOISE_PHILOX if noise is None else L.NOISE_UNIFORM
    if noise is not None:
        assert noise_tokens is not None, 'q_sample_udlm: injected noise needs the uniform tokens too'
        noise = noise.to(x.device, torch.float32).contiguous()
        noise_tokens = noise_tokens.to(x.device, torch.int64).contiguous()
        assert noise.shape == x.shape and noise_tokens.shape == x.shape
    seed, off, row_off = (noise_src.seed, noise_src.next_offset(), noise_src.row_offset) if noise_src else (0, 0, 0)
    out = torch.empty_like(x)
    L.check(lib.sgpo_q_sample_udlm(L.ptr(x), L.ptr(move_chance), B, Lq, int(vocab), noise_mode, L.ptr(noise),
                                   L.ptr(noise_tokens), seed, off, row_off * Lq, L.ptr(out), L.stream_ptr()),
            'sgpo_q_sample_udlm')
    return out


def q_sample_mdlm(x, move_chance, mask_index, noise_src=None, noise=None):
    lib = L.load()
    B, Lq = x.shape
    x = x.contiguous()
    move_chance = move_chance.to(x.device, torch.float32).reshape(-1).contiguous()
    assert move_chance.numel() == B
    noise_mode = L.NOISE_PHILOX if noise is None else L.NOISE_UNIFORM
    if noise is not None:
        noise = noise.to(x.device, torch.float32).contiguous()
    seed, off, row_off = (noise_src.seed, noise_src.next_offset(), noise_src.row_offset) if noise_src else (0, 0, 0)
    out = torch.empty_like(x)
    L.check(lib.sgpo_q_sample_mdlm(L.ptr(x), L.ptr(move_chance), B, Lq, mask_index, noise_mode, L.ptr(noise), seed, off,
                                   row_off * Lq, L.ptr(out), L.stream_ptr()), 'sgpo_q_sample_mdlm')
    return out


def categorical_sample(probs, noise_src=None, noise=None, noise_mode=None):
    lib = L.load()
    K = probs.shape[-1]
    rows = probs.numel() // K
    probs = probs.to(torch.float32).contiguous()
    if not probs.is_cuda:
        raise L.SgpoError('categorical_sample: probabilities must live on the GPU')
    if noise is not None:
        noise = noise.to(probs.device, torch.float32).contiguous()
        noise_mode = L.NOISE_EXPONENTIAL if noise_mode is None else noise_mode
    else:
        noise_mode = L.NOISE_PHILOX
    seed, off, row_off = (noise_src.seed, noise_src.next_offset(), noise_src.row_offset) if noise_src else (0, 0, 0)
    out = torch.empty(probs.shape[:-1], dtype=torch.int64, device=probs.device)
    # the kernel counts rows of K probabilities; the noise source counts sequences (leading dimension)
    row_off = row_off * (rows // probs.shape[0]) if probs.dim() > 1 and probs.shape[0] > 0 else row_off
    L.check(lib.sgpo_categorical_sample(L.ptr(probs), rows, K, noise_mode, L.ptr(noise), seed, off, row_off,
                                        L.ptr(out), L.stream_ptr()), 'sgpo_categorical_sample')
    return out


def gemm_bf16(a, w):
    """Test access: a (M,K) fp32/bf16, w (N,K) -> (M,N) fp32 via the MFMA kernel."""
    lib = L.load()
    a16 = a.to(torch.bfloat16).contiguous()
    w16 = w.to(torch.bfloat16).contiguous()
    M, K = a16.shape
    N = w16.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    L.check(lib.sgpo_gemm_bf16(L.ptr(a16), L.ptr(w16), L.ptr(out), M, N, K, L.stream_ptr()), 'sgpo_gemm_bf16')
    return out


BN_LAYER_ORDER = ['sequence1.0.weight', 'sequence1.0.bias', 'sequence1.2.conv.weight', 'sequence1.2.conv.bias',
                  'sequence1.3.weight', 'sequence1.3.bias', 'conv.weight', 'conv.bias', 'sequence2.0.weight',
                  'sequence2.0.bias', 'sequence2.2.conv.weight', 'sequence2.2.conv.bias']


def bytenet_keys(n_layers):
    """State-dict keys of ByteNetLMTime in blob order (the `network.` prefix of the Lightning checkpoint stripped:
    models/pretraining/pretrained.py:10-15, d3pm_evodiff.py:45,54).  Other entries of a checkpoint (buffers such as the
    timestep encoding table, optimizer or EMA state) are not read."""
    keys = ['embedder.embedder.weight', 'embedder.up_embedder.conv.weight', 'embedder.up_embedder.conv.bias']
    keys += [f'embedder.layers.{n}.{k}' for n in range(n_layers) for k in BN_LAYER_ORDER]
    return keys + ['decoder.conv.weight', 'decoder.conv.bias']


def dit_keys(n_blocks):
    """State-dict keys of the DiT backbone in blob order (`backbone.` prefix stripped)."""
    return DIT_ORDER_HEAD + [f'blocks.{i}.{k}' for i in range(n_blocks) for k in DIT_ORDER_BLOCK] + DIT_ORDER_TAIL


class BytenetDenoiser:
    """ByteNetLMTime forward on the GPU (third-party architecture restated; parity unpinned)."""

    def __init__(self, state_dict, r=128, device=None):
        self.device = init(device)
        self.lib = L.load()
        sd = state_dict
        n_tokens, d_emb = sd['embedder.embedder.weight'].shape
        d_model = sd['embedder.up_embedder.conv.weight'].shape[0]
        d_h, _, ks = sd['embedder.layers.0.conv.weight'].shape
        n_layers = 1 + max(int(k.split('.')[2]) for k in sd if k.startswith('embedder.layers.'))
        self.n_tokens, self.d_model = n_tokens, d_model
        self.cfg = L.BytenetConfig(n_tokens, d_emb, d_model, d_h, n_layers, ks, r)
        blob = _blob(sd, bytenet_keys(n_layers), self.device)
        n = self.lib.sgpo_bytenet_param_count(C.byref(self.cfg))
        if blob.numel() != n:
            raise L.SgpoError(f'ByteNet state dict has {blob.numel()} parameters, library expects {n}')
        h = C.c_void_p()
        L.check(self.lib.sgpo_bytenet_create(C.byref(self.cfg), L.ptr(blob), blob.numel(), C.byref(h)),
                'sgpo_bytenet_create')
        self.handle = h
        self.ws = Workspace(self.device)

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.sgpo_bytenet_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def forward(self, tokens, t, return_hidden=False, out=None):
        """tokens (B,L) int64, t int / (1,) / (B,) int64 timesteps -> logits (B,L,n_tokens) fp32."""
        B, Lq = tokens.shape
        tokens = tokens.contiguous()
        t = torch.as_tensor(t, dtype=torch.int64).reshape(-1).to(self.device).contiguous()
        need = self.lib.sgpo_bytenet_workspace_bytes(self.handle, B, Lq)
        ws = self.ws.get(need)
        base = (ws.data_ptr() + 255) & ~255
        logits = out if out is not None else torch.empty(B, Lq, self.n_tokens, dtype=torch.float32, device=self.device)
        hidden = torch.empty(B, Lq, self.d_model, dtype=torch.float32, device=self.device) if return_hidden else None
        L.check(self.lib.sgpo_bytenet_forward(self.handle, L.ptr(tokens), L.ptr(t), t.numel(), B, Lq, L.ptr(logits),
                                              L.ptr(hidden), C.c_void_p(base), need, L.stream_ptr()),
                'sgpo_bytenet_forward')
        return (logits, hidden) if return_hidden else logits


def reverse_step_d3pm(logits, x, delta_q, q_bar, S, hard=True, noise_src=None, noise=None):
    """D3PM.p_sample given denoiser logits.  noise: fp64 (N,L,S) Exp(1) draws (parity runs)."""
    lib = L.load()
    N, Lq, ld = logits.shape
    logits, x = logits.contiguous(), x.contiguous()
    delta_q = delta_q.to(logits.device, torch.float32).contiguous()
    q_bar = q_bar.to(logits.device, torch.float32).contiguous()
    mode = L.STEP_HARD if hard else L.STEP_SOFT
    if noise is not None:
        noise = noise.to(logits.device, torch.float64).contiguous()
        noise_mode = L.NOISE_EXPONENTIAL
    else:
        noise_mode = L.NOISE_PHILOX
    seed, off, row_off = (noise_src.seed, noise_src.next_offset(), noise_src.row_offset) if (noise_src and hard) else (0, 0, 0)
    x_out = torch.empty_like(x) if hard else None
    probs = None if hard else torch.empty(N, Lq, S, dtype=torch.float64, device=logits.device)
    L.check(lib.sgpo_reverse_step_d3pm(L.ptr(logits), ld, L.ptr(x), N, Lq, S, L.ptr(delta_q), L.ptr(q_bar), mode,
                                       noise_mode, L.ptr(noise), seed, off, row_off * Lq, L.ptr(x_out), L.ptr(probs),
                                       L.stream_ptr()), 'sgpo_reverse_step_d3pm')
    return x_out if hard else probs


def q_sample_d3pm(x, t, q_bar_all, S, noise_src=None, noise=None):
    lib = L.load()
    B, Lq = x.shape
    x = x.contiguous()
    t = torch.as_tensor(t, dtype=torch.int64).reshape(-1).to(x.device).contiguous()
    assert t.numel() == B and q_bar_all.is_cuda and q_bar_all.dtype == torch.float32 and q_bar_all.is_contiguous()
    if noise is not None:
        noise = noise.to(x.device, torch.float32).contiguous()
        noise_mode = L.NOISE_EXPONENTIAL
    else:
        noise_mode = L.NOISE_PHILOX
    seed, off, row_off = (noise_src.seed, noise_src.next_offset(), noise_src.row_offset) if noise_src else (0, 0, 0)
    out = torch.empty_like(x)
    L.check(lib.sgpo_q_sample_d3pm(L.ptr(x), L.ptr(t), L.ptr(q_bar_all), q_bar_all.shape[0], B, Lq, S, noise_mode,
                                   L.ptr(noise), seed, off, row_off * Lq, L.ptr(out), L.stream_ptr()),
            'sgpo_q_sample_d3pm')
    return out


def daps_mh(vf, x0hat, residues, S, log_alpha, log_1alpha, alpha, steps, noise_src=None, r_idx=None, r_val=None,
            r_u=None):
    """DAPS.metropolis_hasting on the GPU.  vf: MlpValueFunction; residues: 0-indexed mutable positions."""
    lib = L.load()
    N, Lq = x0hat.shape
    x0hat = x0hat.contiguous()
    dev = x0hat.device
    residues = [int(r) for r in residues]
    # the kernel indexes x0hat / x_out rows with these positions: refuse anything outside the row here
    if not residues or min(residues) < 0 or max(residues) >= Lq:
        raise L.SgpoError(f'daps_mh: mutable positions {residues} outside a length-{Lq} row')
    # device copies of the position list, its read-slot map and the kernel workspace live ON the value function they
    # were built for (it is rebuilt after every retraining), so nothing outlives it and no other model can pick them up
    cache = vf.__dict__.setdefault('_mh_cache', {})
    key = (tuple(residues), str(dev))
    if key not in cache:
        slot = [vf.read_pos.index(r) if r in vf.read_pos else -1 for r in residues]
        cache[key] = (torch.tensor(residues, dtype=torch.int32, device=dev),
                      torch.tensor(slot, dtype=torch.int32, device=dev),
                      torch.empty(512, dtype=torch.uint8, device=dev))
    d_res, d_slot, ws = cache[key]
    base = (ws.data_ptr() + 255) & ~255
    if r_idx is not None:
        noise_mode = L.NOISE_UNIFORM
        r_idx = r_idx.to(dev, torch.int32).contiguous()
        r_val = r_val.to(dev, torch.int32).contiguous()
        r_u = r_u.to(dev, torch.float32).contiguous()
    else:
        noise_mode = L.NOISE_PHILOX
    seed, off, row_off = (noise_src.seed, noise_src.next_offset(), noise_src.row_offset) if noise_src else (0, 0, 0)
    out = torch.empty_like(x0hat)
    L.check(lib.sgpo_daps_mh(vf.handle, L.ptr(x0hat), N, Lq, S, L.ptr(d_res), L.ptr(d_slot), len(residues),
                             float(log_alpha), float(log_1alpha), float(alpha), int(steps), noise_mode, L.ptr(r_idx),
                             L.ptr(r_val), L.ptr(r_u), seed, off, row_off, L.ptr(out), C.c_void_p(base), 256,
                             L.stream_ptr()), 'sgpo_daps_mh')
    return out
