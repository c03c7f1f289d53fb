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
