"""CPU, world_size 2 over gloo: the N>1 plumbing of the sampling path (shard bounds, ragged
all-gather of tokens + scores, rank ordering).  The shard-invariance of the noise itself is a GPU
property and is tested in tests/test_gpu_kernels.py / test_gpu_sampling.py."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sgpo_amd.dist import gather_samples, shard_bounds, sharded_inference


def test_shard_bounds_partition():
    for n in (1, 7, 10, 4096, 16384):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StubNet:
    length = 13
    global_batch = None

    def reseed(self, seed, row_offset):
        self.seen = (seed, row_offset)


class _StubAlgo:
    """Returns the rows of a fixed 'global batch' that the reseed call selected, like a shard-invariant sampler."""
    device = 'cpu'

    def __init__(self, full):
        self.net, self.full = _StubNet(), full

    def inference(self, num_samples):
        assert num_samples > 0 and self.net.global_batch == self.full.shape[0]
        lo = self.net.seen[1]
        return self.full[lo:lo + num_samples]


def _worker(rank, world, port, n_rows, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(n_rows, rank, world)
        g = torch.Generator().manual_seed(0)
        full = torch.randint(0, 31, (n_rows, 13), generator=g)       # the "global batch" every rank can rebuild
        score = full.float().sum(1)
        t, s = gather_samples(full[lo:hi], score[lo:hi])
        ok = bool(t.dtype == torch.uint8 and torch.equal(t.long(), full) and torch.equal(s, score))
        # the same through sharded_inference with a stub sampler (incl. a rank whose shard is empty: n_rows < world)
        algo = _StubAlgo(full)
        t2, s2 = sharded_inference(algo, n_rows, seed=5, score_fn=lambda x: x.float().sum(1))
        ok = ok and bool(torch.equal(t2.long(), full) and torch.equal(s2, score) and algo.net.seen == (5, lo))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_gather_samples_world2_ragged():
    world = 2
    for n_rows in (10, 7, 1):      # even and ragged shards, and fewer rows than ranks
        port = _free_port()
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, world, port, n_rows, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=120) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
        assert all(ok for _, ok in res), res


def test_gather_samples_single_process():
    x = torch.randint(0, 31, (5, 9))
    t, s = gather_samples(x, None)
    assert t.dtype == torch.uint8 and torch.equal(t.long(), x) and s is None


# ---- work cut by call: the (strength, repeat) grid of config 4 and the Thompson draws of config 5 ----------------
def _stub_cell(cell, key):
    """A 'sampler call' whose rows depend only on (cell, key): what a seeded inference() of that cell returns."""
    coef, rep, b = cell
    g = torch.Generator().manual_seed(key & 0x7FFFFFFFFFFFFFFF)
    x = torch.randint(0, 31, (b, 13), generator=g)
    return x, x.float().sum(1) * coef + rep


def _stub_draw(member, key):
    g = torch.Generator().manual_seed((key ^ (member * 7919)) & 0x7FFFFFFFFFFFFFFF)
    x = torch.randint(0, 4, (6, 5), generator=g)        # few distinct rows: duplicates between draws are common
    return x, x.float().sum(1) + member


class _Store:
    """update_data with BO=True in miniature: unique rows not seen before, truncated to what the round still needs."""

    def __init__(self, n_needed):
        self.seen, self.rows, self.scores, self.n_needed = set(), [], [], n_needed

    def consume(self, t, s):
        new = 0
        for r, v in zip(t.tolist(), s.tolist()):
            if tuple(r) not in self.seen and len(self.rows) < self.n_needed:
                self.seen.add(tuple(r))
                self.rows.append(r)
                self.scores.append(v)
                new += 1
        return new


def _thompson(world, rank, n_needed, max_draws, per_rank=1):
    import numpy as np
    from sgpo_amd.dist import thompson_round
    rng = np.random.RandomState(11)                      # util/seed.py: every rank seeds numpy alike
    store = _Store(n_needed)
    many = (lambda members, keys: [_stub_draw(m, k) for m, k in zip(members, keys)]) if per_rank > 1 else None
    n_new, k = thompson_round(_stub_draw, store.consume, n_needed, max_draws, lambda: int(rng.randint(0, 10)), seed=3,
                              rank=rank, world=world, per_rank=per_rank, draw_many=many)
    return n_new, k, store.rows, store.scores


def _worker_cells(rank, world, port, q):
    from sgpo_amd.dist import run_cells, share_members
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        cells = [(c, r, 3 + (r % 2)) for r in range(3) for c in (0.1, 1.0, 10.0)]       # 9 cells, ragged batch sizes
        got = run_cells(cells, _stub_cell, seed=42)
        want = run_cells(cells, _stub_cell, seed=42, rank=0, world=1)
        ok = len(got) == len(cells) and all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(got, want))
        # ensemble weights trained on different ranks reach every rank
        local = {m: torch.full((5,), float(m)) for m in range(5) if m % world == rank}
        blobs = share_members(local, 5)
        ok = ok and all(torch.equal(b, torch.full((5,), float(m))) for m, b in enumerate(blobs))
        # Thompson round: same accepted rows, same stopping draw as the one-rank loop
        for n_needed, max_draws in ((20, 50), (1000, 7), (3, 50)):
            ok = ok and _thompson(world, rank, n_needed, max_draws) == _thompson(1, 0, n_needed, max_draws)
            # several draws stacked per rank and wave (draw_many): still the one-rank, one-draw-at-a-time result
            ok = ok and _thompson(world, rank, n_needed, max_draws, per_rank=3) == _thompson(1, 0, n_needed, max_draws)
        # -- round 3 --------------------------------------------------------------------------------------------
        # fewer cells / members than ranks: the rank that was dealt nothing contributes nothing, nobody hangs or raises
        one = run_cells(cells[:1], _stub_cell, seed=42)
        ok = ok and len(one) == 1 and torch.equal(one[0][0], want[0][0]) and torch.equal(one[0][1], want[0][1])
        blobs = share_members({0: torch.arange(7.)} if rank == 0 else {}, 1)
        ok = ok and len(blobs) == 1 and torch.equal(blobs[0], torch.arange(7.))
        # a batch costs ONE collective when the row bound is known beforehand; two without it
        calls = []
        real = dist.all_gather_into_tensor
        dist.all_gather_into_tensor = lambda *a, **k: (calls.append(a[0].numel()), real(*a, **k))[1]
        try:
            uniform = [(c, r, 4) for r in range(3) for c in (0.1, 1.0, 10.0)]
            got2 = run_cells(uniform, _stub_cell, seed=42, rows_per_cell=4)
            n_fast = len(calls)
            run_cells(uniform, _stub_cell, seed=42)
            n_ragged = len(calls) - n_fast
            full = torch.randint(0, 31, (10, 13), generator=torch.Generator().manual_seed(0))
            lo, hi = shard_bounds(10, rank, world)
            before = len(calls)
            t, s_ = gather_samples(full[lo:hi], full[lo:hi].float().sum(1), max_rows=5)
            n_rows_call = len(calls) - before
        finally:
            dist.all_gather_into_tensor = real
        want2 = run_cells(uniform, _stub_cell, seed=42, rank=0, world=1)
        ok = ok and all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(got2, want2))
        ok = ok and (n_fast, n_ragged, n_rows_call) == (1, 2, 1) and torch.equal(t.long(), full)
        # Thompson picks from their own stream (member_stream): draws that use numpy's GLOBAL generator in between (the
        # reference's project_sequences does) leave the picks, and with them the round, independent of the wave size
        from sgpo_amd.dist import member_stream, thompson_round
        import numpy as np

        def noisy_draw(member, key):
            np.random.rand(3)                           # what string repair does to the global generator
            return _stub_draw(member, key)
        res = []
        for w, r, pr in ((world, rank, 1), (world, rank, 3), (1, 0, 1)):
            np.random.seed(5)
            st = _Store(20)
            many = (lambda ms, ks: [noisy_draw(m, k) for m, k in zip(ms, ks)]) if pr > 1 else None
            res.append((thompson_round(noisy_draw, st.consume, 20, 50, member_stream(11, 10), seed=3, rank=r, world=w,
                                       per_rank=pr, draw_many=many, rows_per_draw=6), st.rows))
        ok = ok and res[0] == res[2] and res[1] == res[2]
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_cells_and_thompson_draws_world2_equal_the_single_rank_run():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_cells, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_item_seed_and_deal():
    from sgpo_amd.dist import deal, item_seed
    assert sorted(sum((deal(10, r, 4) for r in range(4)), [])) == list(range(10))
    assert deal(2, 3, 4) == []
    keys = {item_seed(42, i) for i in range(1000)} | {item_seed(43, i) for i in range(1000)}
    assert len(keys) == 2000 and all(0 <= k < 2 ** 64 for k in keys)
