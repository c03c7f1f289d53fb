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


def _wire(group):
    """Device the collective runs on: gloo carries host tensors only (a rehearsal of the N > 1 flow with GPU tensors
    stages the exchange through the host), RCCL takes the GPU tensors where they are."""
    return torch.device('cpu') if dist.get_backend(group) == 'gloo' else None


@torch.no_grad()
def gather_rows(tokens, scores=None, tags=None, max_rows=None, group=None):
    """ONE all-gather of per-rank sample rows: every row travels as L token bytes + 4 score
    bytes + 8 tag bytes in one padded uint8 matrix, the rank's row count in a 16-byte header in front of it.

    tokens (b_r, L) integer (< 256: the 31-letter alphabet), scores (b_r,) fp32 or None, tags (b_r,) int64 or None.
    max_rows: an upper bound of b_r that every rank knows without asking (a shard size, draws x batch): with it the
    exchange is exactly one collective.  Without it the ranks first exchange (b_r, L) -- a second, 16-byte
    collective -- which is also how a rank that produced nothing learns the row width (tokens may then be None).
    Returns (tokens uint8 (sum b_r, L), scores or None, tags or None) in rank order on every rank."""
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    if world == 1:
        return tokens.to(torch.uint8), scores, tags
    home = tokens.device if tokens is not None else None
    dev = _wire(group) or home
    n = 0 if tokens is None else int(tokens.shape[0])
    Lq = 0 if tokens is None else int(tokens.shape[1])
    if max_rows is None or tokens is None:
        mine = torch.tensor([n, Lq], dtype=torch.int64, device=dev or 'cpu')
        sizes = torch.empty(world * 2, dtype=torch.int64, device=mine.device)
        dist.all_gather_into_tensor(sizes, mine, group=group)
        sizes = sizes.view(world, 2).cpu()
        Lq = int(sizes[:, 1].max())
        max_rows = max(int(sizes[:, 0].max()), 1 if max_rows is None else max_rows)
        if tokens is None:
            home = home or mine.device
            dev = dev or home
            tokens = torch.empty(0, Lq, dtype=torch.uint8, device=dev)
    assert n <= max_rows, f'gather_rows: {n} rows exceed max_rows {max_rows}'
    width = Lq + 12
    buf = torch.zeros(16 + max_rows * width, dtype=torch.uint8, device=dev)
    flags = (1 if scores is not None else 0) | (2 if tags is not None else 0)
    buf[:16] = torch.tensor([n, flags], dtype=torch.int64).view(torch.uint8).to(dev)
    body = buf[16:].view(max_rows, width)
    if n:
        body[:n, :Lq] = tokens.detach().to(dev).to(torch.uint8)
        if scores is not None:
            body[:n, Lq:Lq + 4] = scores.detach().to(dev, torch.float32).contiguous().view(torch.uint8).view(n, 4)
        if tags is not None:
            body[:n, Lq + 4:] = tags.detach().to(dev, torch.int64).contiguous().view(torch.uint8).view(n, 8)
    out = torch.empty(world * buf.numel(), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, buf, group=group)
    out = out.view(world, buf.numel())
    head = out[:, :16].flatten().clone().view(torch.int64).view(world, 2).cpu()          # (world, 2): count, flags -- one small read
    counts = [int(c) for c in head[:, 0]]
    any_flags = 0
    for c, f in zip(counts, head[:, 1].tolist()):
        if c:
            any_flags |= int(f)
    bodies = out[:, 16:].view(world, max_rows, width)
    rows = torch.cat([bodies[r, :c] for r, c in enumerate(counts)]) if sum(counts) else bodies[0, :0]
    nrow = rows.shape[0]
    out_t = rows[:, :Lq].contiguous().to(home)
    out_s = rows[:, Lq:Lq + 4].flatten().clone().view(torch.float32).view(nrow).to(home) if any_flags & 1 else None
    out_g = rows[:, Lq + 4:].flatten().clone().view(torch.int64).view(nrow).to(home) if any_flags & 2 else None
    return out_t, out_s, out_g


def gather_samples(tokens, scores=None, group=None, max_rows=None):
    """All-gather per-rank samples: tokens (b_r, L) integer -> (sum b_r, L) uint8 on every rank, scores (b_r,) fp32 ->
    (sum b_r,).  Ranks may hold different row counts.  One collective when `max_rows` is given (`gather_rows`)."""
    t, s, _ = gather_rows(tokens, scores, None, max_rows=max_rows, group=group)
    return t, s


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
    return gather_samples(x, s, max_rows=-(-num_samples // world))       # ceil: every rank knows it, one collective


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


def gather_tagged(tokens, scores, tags, group=None, max_rows=None):
    """One ragged all-gather of rows that carry an item tag: tokens (n, L) (None: this rank has no rows), scores (n,)
    fp32 or None, tags (n,) int64.  Returns the rows of all ranks grouped by tag, original order inside a tag:
    {tag: (tokens uint8, scores)}.  Tokens, scores and tags share the collective (`gather_rows`)."""
    t, s, all_tags = gather_rows(tokens, scores, tags, max_rows=max_rows, group=group)
    out = {}
    if all_tags is None:
        return out
    for tag in sorted(set(all_tags.tolist())):
        rows = torch.nonzero(all_tags == tag).squeeze(-1)
        out[int(tag)] = (t[rows], None if s is None else s[rows])
    return out


def run_cells(cells, run_cell, seed, rank=None, world=None, group=None, rows_per_cell=None):
    """Independent sampler calls dealt over the ranks (config 4: the (guidance strength, repeat) grid of pareto.py:131-164).

    cells: list of picklable descriptions, the same list on every rank.  run_cell(cell, key) -> (tokens (b, L) int,
    scores (b,) or None) runs ONE cell whole on the calling rank with `key = item_seed(seed, index)` as its Philox key
    (so a cell's samples do not depend on which rank, or how many ranks, ran the job).  Returns, on every rank, the
    list of (tokens uint8, scores) in cell order after ONE all-gather of everything this rank produced.
    rows_per_cell: the rows a cell yields when every cell yields the same (pareto.py: 200) -- makes the exchange a
    single collective.  More ranks than cells is fine: a rank that was dealt nothing contributes no rows."""
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
    max_rows = None if rows_per_cell is None else rows_per_cell * (-(-len(cells) // world))
    merged = gather_tagged(torch.cat(toks) if toks else None, torch.cat(scs) if (toks and have_scores) else None,
                           torch.cat(tags) if toks else None, group=group, max_rows=max_rows if toks else None)
    return [merged[i] for i in range(len(cells))]


@torch.no_grad()
def share_members(local, n_members, rank=None, world=None, group=None):
    """Ensemble weights after sharded training (config 5): member m was trained on rank m % world and `local[m]` holds
    its flat fp32 blob there.  Returns the list of all n_members blobs on every rank (one broadcast per member from
    its owner: 2 MB each for the CreiLOV value function)."""
    rank, world = _world(rank, world, group)
    if world == 1:
        return [local[m] for m in range(n_members)]
    # a rank that owns no member (n_members < world) learns blob size and device from member 0's owner, rank 0
    gloo = dist.get_backend(group) == 'gloo'
    dev0 = next(iter(local.values())).device if local else torch.device('cpu' if (gloo or not torch.cuda.is_available()) else 'cuda')
    numel = torch.tensor([local[0].numel() if 0 in local else 0], dtype=torch.int64, device='cpu' if gloo else dev0)
    dist.broadcast(numel, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
    out = []
    for m in range(n_members):
        owner = m % world
        buf = local[m].contiguous() if owner == rank else torch.empty(int(numel.item()), dtype=torch.float32, device=dev0)
        wire = buf.cpu() if (buf.is_cuda and dist.get_backend(group) == 'gloo') else buf
        dist.broadcast(wire, src=owner if group is None else dist.get_global_rank(group, owner), group=group)
        out.append(buf if wire is buf else wire.to(buf.device))
    return out


def member_stream(seed, n_members):
    """`pick_member` for `thompson_round`: Thompson picks from a numpy RandomState of their own (seeded like the round,
    iterativeBO.py:202), independent of the global generator that project_sequences draws from."""
    import numpy as np
    rng = np.random.RandomState(int(seed) & 0xFFFFFFFF)
    return lambda: int(rng.randint(0, n_members))


def thompson_round(draw, consume, n_needed, max_draws, pick_member, seed, rank=None, world=None, group=None,
                   per_rank=1, draw_many=None, rows_per_draw=None):
    """One acquisition round of Thompson sampling over an ensemble (iterativeBO.py:204-237) on `world` ranks.

    The reference loops: pick a member (numpy RNG), sample a small batch guided by it, hand the batch to update_data,
    stop when the round has its new sequences or after max_iterations draws.  Draws are independent given
    (member, key), so they are computed a wave of `world * per_rank` at a time -- rank r computes draws
    wave[r * per_rank : (r + 1) * per_rank] -- and all-gathered once per wave; every rank then CONSUMES the wave in
    draw order and stops at the same draw the one-rank loop would have stopped at (draws of the wave beyond that point
    are dropped unseen).
      pick_member() -> member index; called once per draw, in draw order, on EVERY rank, for a whole wave BEFORE any
                       of its draws is computed or consumed.  For results that do not depend on `world` / `per_rank` it
                       must therefore draw from a stream of its OWN (`member_stream(seed)` below): the reference
                       interleaves `np.random.randint` with `inference(detokenize=True)` -> project_sequences, whose
                       np.random.choice draws share numpy's GLOBAL generator (iterativeBO.py:213-234, sampling/base.py:
                       44-63), so with the global generator the picks would shift with the wave size.  String repair
                       keeps the global generator; the picks get their own;
      draw(member, key) -> (tokens (b, L), scores (b,) or None), this rank's sampler call;
      draw_many(members, keys) -> list of such pairs: per_rank draws stacked into ONE sampler call
                       (`Classifier_Guidance_Inpaint.inference_many`: one denoiser forward per step for all of them);
      consume(tokens uint8, scores) -> number of new sequences accepted (update_data on every rank, or on rank 0
                       with the count broadcast by the caller).
      rows_per_draw: the sequences one draw yields (iterativeBO.py:130: 10) -- makes a wave's exchange one collective.
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
            got = gather_tagged(toks, scs, tag, group=group, max_rows=rows_per_draw * per_rank if rows_per_draw else None)
        for d in wave:
            if n_new >= n_needed or d >= max_draws:
                break
            t, sc = got[d]
            n_new += int(consume(t, sc))
            k = d + 1
        else:
            k = wave[-1] + 1
    return n_new, k
