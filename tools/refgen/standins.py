"""In-process stand-ins for the third-party packages the reference imports but
this container lacks.  Used ONLY by tools/refgen/make_goldens.py
to import /root/reference on CPU and capture golden vectors.  Nothing here is
product code and nothing here travels into the tests, bench or smoke paths.

Shell stand-ins carry no arithmetic.  Two carry arithmetic and are therefore
*definitions* (documented in DESIGN.md):
  * flash_attn  -> softmax(q k^T / sqrt(d)) v via F.scaled_dot_product_attention,
                   rotate-half RoPE on q,k with cos/sin (L, d/2)
  * evodiff.utils.Tokenizer -> alphabet of configs/data/*.yaml:10, K=26, mask 28
"""
import importlib.machinery
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ALPHABET = 'ACDEFGHIKLMNPQRSTVWYBZXJOU-*#@!'


def _mod(name):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__path__ = []
    sys.modules[name] = m
    parent, _, child = name.rpartition('.')
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(d):
    if isinstance(d, dict):
        return _AttrDict({k: to_attr(v) for k, v in d.items()})
    return d


class Tokenizer:
    """Stand-in for evodiff.utils.Tokenizer(sequences=True)."""

    def __init__(self, sequences=True, **kw):
        self.alphabet = list(ALPHABET)
        self.a_to_i = {a: i for i, a in enumerate(self.alphabet)}
        self.i_to_a = np.array(self.alphabet)
        self.K = 26
        self.mask_id = 28
        self.pad_id = 30

    def tokenize(self, seq):
        return np.array([self.a_to_i[a] for a in seq[0]])

    def untokenize(self, x):
        return ''.join(self.i_to_a[int(t)] for t in x)

    def q_random_schedule(self, timesteps=500):
        # sohl-dickstein schedule, fp64 cumulative product cast to fp32
        K = self.K
        betas = 1.0 / (timesteps - np.arange(1, timesteps + 1) + 1)
        Q_prod = torch.eye(K, dtype=torch.float64)
        q_ts, q_prods = [], []
        for b in betas:
            q = torch.full((K, K), b / K, dtype=torch.float64) + torch.eye(K, dtype=torch.float64) * (1 - b)
            Q_prod = Q_prod @ q
            q_ts.append(q.float())
            q_prods.append(Q_prod.float())
        return torch.stack(q_prods), torch.stack(q_ts)


def install():
    # --- lightning / pytorch_lightning ---------------------------------
    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device('cpu')

        @property
        def dtype(self):
            return torch.float32

    for base in ('pytorch_lightning', 'lightning'):
        m = _mod(base)
        m.LightningModule = LightningModule
        m.Trainer = object
        m.seed_everything = lambda *a, **k: None
        cb = _mod(base + '.callbacks')
        cb.Callback = object
        cb.ModelCheckpoint = object
        cb.LearningRateMonitor = object
        lg = _mod(base + '.loggers')
        lg.WandbLogger = object
    lp = _mod('lightning.pytorch')
    lp.LightningModule = LightningModule
    ut = _mod('lightning.pytorch.utilities')
    ut.rank_zero_only = lambda f: f

    # --- torchmetrics ----------------------------------------------------
    tm = _mod('torchmetrics')

    class Metric(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    class MetricCollection(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def set_dtype(self, *a):
            pass

        def clone(self, *a, **k):
            return MetricCollection()

    tm.Metric = Metric
    tm.MetricCollection = MetricCollection
    ag = _mod('torchmetrics.aggregation')
    ag.MeanMetric = Metric

    # --- hydra / omegaconf ----------------------------------------------
    hy = _mod('hydra')
    hu = _mod('hydra.utils')

    def instantiate(cfg, *a, **k):
        if isinstance(cfg, dict) and '_target_' in cfg:
            tgt = cfg['_target_']
            if tgt.endswith('Tokenizer'):
                return Tokenizer(**k)
            raise NotImplementedError(tgt)
        if callable(cfg):
            return cfg(*a, **k)
        raise NotImplementedError(cfg)

    hu.instantiate = instantiate
    hy.main = lambda *a, **k: (lambda f: f)
    oc = _mod('omegaconf')

    class OmegaConf:
        @staticmethod
        def create(d):
            return to_attr(d)

    oc.OmegaConf = OmegaConf
    oc.DictConfig = dict
    oce = _mod('omegaconf.errors')
    oce.ConfigAttributeError = AttributeError

    # --- misc shells -------------------------------------------------------
    timm = _mod('timm')
    ts = _mod('timm.scheduler')
    ts.CosineLRScheduler = type('CosineLRScheduler', (), {})
    _mod('wandb')
    _mod('gpytorch')
    bo = _mod('botorch')
    bm = _mod('botorch.models')
    bm.SingleTaskGP = object
    _mod('esm')
    _mod('Bio')

    # --- flash_attn (DEFINITION) -----------------------------------------
    fa = _mod('flash_attn')
    fl = _mod('flash_attn.layers')
    fr = _mod('flash_attn.layers.rotary')
    fi = _mod('flash_attn.flash_attn_interface')

    def apply_rotary_emb_qkv_(qkv, cos, sin):
        # qkv (B, L, 3, H, D); cos/sin (L, D/2); rotate-half over the full head dim on q, k
        d2 = cos.shape[-1]
        c = cos[None, :, None, :]
        s = sin[None, :, None, :]
        for i in (0, 1):
            x = qkv[:, :, i]
            x1, x2 = x[..., :d2].clone(), x[..., d2:2 * d2].clone()
            x[..., :d2] = x1 * c - x2 * s
            x[..., d2:2 * d2] = x1 * s + x2 * c
        return qkv

    fr.apply_rotary_emb_qkv_ = apply_rotary_emb_qkv_

    def flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, p, causal=False):
        # qkv (T, 3, H, D), all sequences the same length max_seqlen
        T, _, H, D = qkv.shape
        B = T // max_seqlen
        q, k, v = (qkv[:, i].reshape(B, max_seqlen, H, D).transpose(1, 2) for i in range(3))
        o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
        return o.transpose(1, 2).reshape(T, H, D)

    fi.flash_attn_varlen_qkvpacked_func = flash_attn_varlen_qkvpacked_func
    fa.flash_attn_interface = fi
    fa.layers = fl

    # --- evodiff / sequence_models ----------------------------------------
    ev = _mod('evodiff')
    eu = _mod('evodiff.utils')
    eu.Tokenizer = Tokenizer
    ep = _mod('evodiff.pretrained')
    ep.D3PM_UNIFORM_38M = None
    el = _mod('evodiff.losses')
    el.D3PMCELoss = object
    el.D3PMLVBLoss = object
    ec = _mod('evodiff.collaters')
    ec.D3PMCollater = object
    _mod('evodiff.model')
    sm = _mod('sequence_models')
    smm = _mod('sequence_models.metrics')
    smm.MaskedAccuracy = object
    smc = _mod('sequence_models.constants')
    smc.MSA_ALPHABET = ALPHABET
    return True
