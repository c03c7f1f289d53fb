"""Multi-GPU sharding of the sampling path.

Every sequence of a batch is independent in every sampler, so a batch shards over ranks with NO
data-path collective: rank r samples rows [lo, hi) with Philox counters offset by `lo`
(shard-invariant noise).  The only exchange is one all-gather per batch of the generated tokens
(uint8) and their scores (fp32) -- RCCL over xGMI on GPUs (`backend='nccl'`), gloo on CPU for tests.
One process per GPU; launch with torch.distributed.run.

Three ways to cut the work, all ending in that one all-gather:
  * `sharded_inference`  -- rows of ONE batch over the ranks (BASELINE config 3: 16384 = 8 x 2048);
  * `run_cells`          -- independent sampler calls, one (guidance strength, repeat) cell each, dealt to the
                            ranks (config 4: pareto.py:131-164 loops over them one after the other);
  * `thompson_round`     -- Thompson draws of an ensemble (config 5: iterativeBO.py:204-237), a wave of
                            `world` draws at a time, consumed in draw order so the result equals the
                            one-rank loop; `share_members` hands every rank the weights of the members
                            that were trained elsewhere.
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


@torch.no_grad()
def gather_samples(tokens, scores=None, group=None):
    """All-gather per-rank samples.  tokens (b_r, L) integer -> (sum b_r, L) uint8 on every rank
    (tokens < 256: the 31-letter alphabet), scores (b_r,) fp32 -> (sum b_r,).  Ranks may hold
    different row counts (ragged shards are padded to the maximum and trimmed)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tokens.to(torch.uint8), scores
    world = dist.get_world_size(group)
    home = tokens.device
    # gloo carries host tensors only: a rehearsal of the N > 1 flow with GPU tensors stages the exchange through the host
    dev = torch.device('cpu') if dist.get_backend(group) == 'gloo' else home
    tokens = tokens.detach().to(dev)
    scores = scores if scores is None else scores.detach().to(dev)
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
    out_t = all_t[keep].to(home)
    out_s = None
    if scores is not None:
        pad_s = torch.zeros(mx, dtype=torch.float32, device=dev)
        pad_s[:scores.shape[0]] = scores.to(torch.float32)
        all_s = torch.empty(world * mx, dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(all_s, pad_s, group=group)
        out_s = all_s[keep].to(home)
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


# ---------------------------------------------------------------------------------------------------------
# Work that is cut by CALL instead of by row (configs 4 and 5)

def _world(rank, world, group=None):
    if world is None:
        on = dist.is_available() and dist.is_initialized()
        return (dist.get_rank(group) if on else 0), (dist.get_world_size(group) if on else 1)
    return rank, world


def deal(n_items, rank, world):
    """Items [0, n_items) dealt round-robin: rank r takes r, r + world, ...  (cells / draws cost the same, so the
    deal is balanced to within one item)."""
    return list(range(rank, n_items, world))


def item_seed(seed, index):
    """Philox key of item `index` of a job keyed by `seed`: the same on whichever rank runs the item, different
    for every item (splitmix64 of the pair)."""
    z = (int(seed) * 0x9E3779B97F4A7C15 + int(index) + 1) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def gather_tagged(tokens, scores, tags, group=None):
    """One ragged all-gather of rows that carry an item tag: tokens (n, L), scores (n,) fp32 or None, tags (n,)
    int64.  Returns the rows of all ranks sorted by (tag, original order): {tag: (tokens uint8, scores)}."""
    t, s = gather_samples(tokens, scores, group=group)
    # the tags ride along as a second byte matrix (8 bytes per row) in the same padded layout
    tag_bytes = tags.to(torch.int64).contiguous().view(torch.uint8).reshape(-1, 8)
    g, _ = gather_samples(tag_bytes, None, group=group)
    all_tags = g.contiguous().view(torch.int64).reshape(-1)
    out = {}
    for tag in sorted(set(all_tags.tolist())):
        rows = torch.nonzero(all_tags == tag).squeeze(-1)
        out[int(tag)] = (t[rows], None if s is None else s[rows])
    return out


def run_cells(cells, run_cell, seed, rank=None, world=None, group=None):
    """Independent sampler calls dealt over the ranks (config 4: the (guidance strength, repeat) grid of pareto.py:131-164).

    cells: list of picklable descriptions, the same list on every rank.  run_cell(cell, key) -> (tokens (b, L) int,
    scores (b,) or None) runs ONE cell whole on the calling rank with `key = item_seed(seed, index)` as its Philox key
    (so a cell's samples do not depend on which rank, or how many ranks, ran the job).  Returns, on every rank, the
    list of (tokens uint8, scores) in cell order after ONE all-gather of everything this rank produced."""
    rank, world = _world(rank, world, group)
    mine = deal(len(cells), rank, world)
    toks, scs, tags = [], [], []
    have_scores = None
    for i in mine:
        x, sc = run_cell(cells[i], item_seed(seed, i))
        toks.append(x)
        tags.append(torch.full((x.shape[0],), i, dtype=torch.int64, device=x.device))
        have_scores = sc is not None if have_scores is None else have_scores
        if sc is not None:
            scs.append(sc.to(torch.float32))
    if world == 1:
        return [(toks[k].to(torch.uint8), scs[k] if have_scores else None) for k in range(len(cells))]
    if not toks:        # more ranks than cells: an empty contribution (the collective still needs every rank)
        raise ValueError('run_cells: every rank needs at least one cell (len(cells) >= world); run fewer ranks')
    merged = gather_tagged(torch.cat(toks), torch.cat(scs) if have_scores else None, torch.cat(tags), group=group)
    return [merged[i] for i in range(len(cells))]


@torch.no_grad()
def share_members(local, n_members, rank=None, world=None, group=None):
    """Ensemble weights after sharded training (config 5): member m was trained on rank m % world and `local[m]` holds
    its flat fp32 blob there.  Returns the list of all n_members blobs on every rank (one broadcast per member from
    its owner: 2 MB each for the CreiLOV value function)."""
    rank, world = _world(rank, world, group)
    if world == 1:
        return [local[m] for m in range(n_members)]
    ref = next(iter(local.values()))
    out = []
    for m in range(n_members):
        owner = m % world
        buf = local[m].contiguous() if owner == rank else torch.empty_like(ref)
        wire = buf.cpu() if (buf.is_cuda and dist.get_backend(group) == 'gloo') else buf
        dist.broadcast(wire, src=owner if group is None else dist.get_global_rank(group, owner), group=group)
        out.append(buf if wire is buf else wire.to(buf.device))
    return out


def thompson_round(draw, consume, n_needed, max_draws, pick_member, seed, rank=None, world=None, group=None,
                   per_rank=1, draw_many=None):
    """One acquisition round of Thompson sampling over an ensemble (iterativeBO.py:204-237) on `world` ranks.

    The reference loops: pick a member (numpy RNG), sample a small batch guided by it, hand the batch to update_data,
    stop when the round has its new sequences or after max_iterations draws.  Draws are independent given
    (member, key), so they are computed a wave of `world * per_rank` at a time -- rank r computes draws
    wave[r * per_rank : (r + 1) * per_rank] -- and all-gathered once per wave; every rank then CONSUMES the wave in
    draw order and stops at the same draw the one-rank loop would have stopped at (draws of the wave beyond that point
    are dropped unseen).
      pick_member() -> member index; called once per draw, in draw order, on EVERY rank (same numpy seed on all
                       ranks, as util/seed.py sets it, keeps the streams in step);
      draw(member, key) -> (tokens (b, L), scores (b,) or None), this rank's sampler call;
      draw_many(members, keys) -> list of such pairs: per_rank draws stacked into ONE sampler call
                       (`Classifier_Guidance_Inpaint.inference_many`: one denoiser forward per step for all of them);
      consume(tokens uint8, scores) -> number of new sequences accepted (update_data on every rank, or on rank 0
                       with the count broadcast by the caller).
    Returns (n_new, n_draws_consumed)."""
    rank, world = _world(rank, world, group)
    n_new = k = 0
    while n_new < n_needed and k < max_draws:
        # a full wave is always computed, also when fewer draws remain before max_draws: the surplus is dropped below
        # (pick_member is thereby called up to wave - 1 times more than in the one-rank loop; the reference reseeds numpy
        # at the start of every round, iterativeBO.py:202, so rounds stay reproducible)
        wave = list(range(k, k + world * per_rank))
        members = [pick_member() for _ in wave]
        mine = range(rank * per_rank, (rank + 1) * per_rank)
        keys = [item_seed(seed, wave[j]) for j in mine]
        if draw_many is not None:
            outs = draw_many([members[j] for j in mine], keys)
        else:
            outs = [draw(members[j], key) for j, key in zip(mine, keys)]
        if world == 1:
            got = {wave[j]: (x.to(torch.uint8), sc) for j, (x, sc) in zip(mine, outs)}
        else:
            toks = torch.cat([x for x, _ in outs])
            have = outs[0][1] is not None
            scs = torch.cat([sc.to(torch.float32) for _, sc in outs]) if have else None
            tag = torch.cat([torch.full((x.shape[0],), wave[j], dtype=torch.int64, device=x.device) for j, (x, _) in zip(mine, outs)])
            got = gather_tagged(toks, scs, tag, group=group)
        for d in wave:
            if n_new >= n_needed or d >= max_draws:
                break
            t, sc = got[d]
            n_new += int(consume(t, sc))
            k = d + 1
        else:
            k = wave[-1] + 1
    return n_new, k
