"""World-2 jobs with the REAL sampler on the GPU (gloo between two processes that share the box's one GPU): the gathered
result equals the one-process run bit for bit.  tests/test_dist_gloo.py covers the same plumbing on CPU with a stub sampler;
the emulated-rank tests of test_gpu_sampling.py cover shard invariance inside one process.  RCCL itself needs one GPU per rank
and only runs in the driver's 8-GPU bench."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _launch(mode, out_path, world=2):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dist_gpu_worker.py'), mode, out_path], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=100)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode(errors='replace')[-3000:])
    assert all(p.returncode == 0 for p in procs), '\n----\n'.join(logs)
    return torch.load(out_path, weights_only=True)


@pytest.fixture(scope='module')
def parts():
    import bench
    net, pred, algo, sd, msd = bench.build('cuda:0', 301)
    net.timestep = 9
    return net, pred, algo


def test_two_ranks_row_sharded_batch_equals_one_rank(parts, tmp_path):
    net, pred, algo = parts
    got = _launch('rows', str(tmp_path / 'rows.pt'))
    net.reseed(seed=5, row_offset=0)
    whole = algo.inference(num_samples=301)
    assert got['tokens'].dtype == torch.uint8 and torch.equal(got['tokens'].long(), whole.cpu())
    assert torch.equal(got['scores'], pred.log_likelihood(whole).detach().float().cpu())


def test_two_ranks_dealt_cells_equal_one_rank(parts, tmp_path):
    from sgpo_amd import dist as D
    net, pred, algo = parts
    got = _launch('cells', str(tmp_path / 'cells.pt'))
    cells = [10, 7, 10, 3, 10]
    for i, n in enumerate(cells):
        net.reseed(seed=D.item_seed(11, i), row_offset=0)
        x = algo.inference(num_samples=n)
        assert torch.equal(got['tokens'][i].long(), x.cpu()), i
        assert torch.equal(got['scores'][i], pred.log_likelihood(x).detach().float().cpu()), i


def test_two_ranks_stacked_thompson_draws_equal_the_one_rank_loop(tmp_path):
    """Config 5 across ranks with the real sampler: waves of 2 ranks x 2 stacked draws, consumed in draw order, stop at the
    same draw with the same tokens and scores as one rank drawing one at a time."""
    sys.path.insert(0, HERE)
    import bench
    import dist_gpu_worker as worker
    from sgpo_amd import dist as D
    got = _launch('thompson', str(tmp_path / 'thompson.pt'))
    want = worker.thompson_job(D, bench, 0, 1, per_rank=1)
    assert got['result'] == want['result'] == [18, 6]
    assert len(got['tokens']) == len(want['tokens']) == 6
    for a, b in zip(got['tokens'], want['tokens']):
        assert torch.equal(a, b)
    for a, b in zip(got['scores'], want['scores']):
        assert torch.equal(a, b)
