"""Multi-GPU sharding of the sampling path.

Every sequence of a batch is independent in every sampler, so a batch shards over ranks with NO
data-path collective: rank r samples rows [lo, hi) with Philox counters offset by `lo`
(shard-invariant noise).  The only exchange is one all-gather per batch of the generated tokens
(uint8) and their scores (fp32) -- RCCL over xGMI on GPUs (`backend='nccl'`), gloo on CPU for tests.
One process per GPU; launch with torch.distributed.run.
"""
import os

import torch
import torch.distributed as dist


def init_process_group(backend=None, device=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / MASTER_*). Returns (rank, world)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and not dist.is_initialized():
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        kw = {}
        if backend == 'nccl' and device is not None:
            kw['device_id'] = torch.device(device)
        dist.init_process_group(backend, **kw)
    return rank, world


def shard_bounds(n_rows, rank, world):
    """Contiguous, balanced split of n_rows over `world` ranks: rows [lo, hi) of rank `rank`."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_samples(tokens, scores=None, group=None):
    """All-gather per-rank samples.  tokens (b_r, L) integer -> (sum b_r, L) uint8 on every rank
    (tokens < 256: the 31-letter alphabet), scores (b_r,) fp32 -> (sum b_r,).  Ranks may hold
    different row counts (ragged shards are padded to the maximum and trimmed)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tokens.to(torch.uint8), scores
    world = dist.get_world_size(group)
    dev = tokens.device
    n = torch.tensor([tokens.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    Lq = tokens.shape[1]
    pad_t = torch.zeros(mx, Lq, dtype=torch.uint8, device=dev)
    pad_t[:tokens.shape[0]] = tokens.to(torch.uint8)
    all_t = torch.empty(world * mx, Lq, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(all_t, pad_t, group=group)
    keep = torch.cat([torch.arange(r * mx, r * mx + c, device=dev) for r, c in enumerate(counts)])
    out_t = all_t[keep]
    out_s = None
    if scores is not None:
        pad_s = torch.zeros(mx, dtype=torch.float32, device=dev)
        pad_s[:scores.shape[0]] = scores.to(torch.float32)
        all_s = torch.empty(world * mx, dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(all_s, pad_s, group=group)
        out_s = all_s[keep]
    return out_t, out_s


def sharded_inference(algo, num_samples, seed, rank=None, world=None, score_fn=None, **kw):
    """Run `algo.inference` on this rank's shard of a `num_samples` batch and gather the whole batch.

    The Philox key is `seed` and the counter offset is the shard's first global row, so the
    gathered batch is identical to a single-rank run of the same seed (tests/test_gpu_sampling.py).
    """
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(num_samples, rank, world)
    algo.net.reseed(seed=seed, row_offset=lo)
    if hasattr(algo.net, 'global_batch'):
        algo.net.global_batch = num_samples          # NOS: KLDivLoss(batchmean) divides by the unsharded batch
    try:
        if hi > lo:
            x = algo.inference(num_samples=hi - lo, **kw)
        else:   # more ranks than samples: this rank contributes no rows (the kernels take no empty batch)
            x = torch.empty(0, int(algo.net.length), dtype=torch.int64, device=getattr(algo, 'device', None) or 'cpu')
    finally:
        if hasattr(algo.net, 'global_batch'):
            algo.net.global_batch = None
    s = None
    if score_fn is not None:
        s = score_fn(x) if hi > lo else torch.empty(0, dtype=torch.float32, device=x.device)
    return gather_samples(x, s)
