"""Rank body of tests/test_gpu_dist.py (not collected by pytest): one process of a world-2 job on ONE GPU over gloo.
usage: dist_gpu_worker.py <mode> <out.pt>   with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def thompson_job(D, bench, rank, world, per_rank):
    """A Thompson round with 4 synthetic ensemble members (the same on every rank); returns what `consume` saw."""
    import numpy as np
    from sgpo_amd import weights as W
    net, pred, algo, sd, msd = bench.build('cuda:0', 10)     # (puts the drop-in packages on sys.path)
    from problem.protein import ProteinPredictor
    net.timestep = 9
    dc = dict(alphabet=bench.ALPHABET, alphabet_size=31, seq_len=119, full_seq=bench.CREILOV, residues=None)
    members = []
    for m in range(4):
        p = ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0')
        p.model.load_state_dict(W.mlp_synthetic_state(seed=300 + m, n_residues=119))
        members.append(p.eval())
    rng = np.random.RandomState(13)
    seen = []

    def draw(m, key):
        algo.update_model(members[m])
        net.reseed(seed=key, row_offset=0)
        x = algo.inference(num_samples=10)
        return x, members[m].log_likelihood(x)

    def draw_many(ms, keys):
        xs = algo.inference_many([members[m] for m in ms], keys, num_samples=10)
        return [(x, members[m].log_likelihood(x)) for m, x in zip(ms, xs)]

    def consume(t, s):
        seen.append((t.cpu(), s.detach().cpu()))
        return 3
    res = D.thompson_round(draw, consume, n_needed=16, max_draws=40, pick_member=lambda: int(rng.randint(0, 4)), seed=31,
                           rank=rank, world=world, per_rank=per_rank, draw_many=draw_many if per_rank > 1 else None)
    return dict(result=list(res), tokens=[t for t, _ in seen], scores=[s for _, s in seen])


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)                      # every rank on the box's one GPU (RCCL would refuse that; gloo does not care)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bench
    from sgpo_amd import dist as D
    out = {}
    try:
        if mode == 'rows':          # one batch, rows sharded (ragged: 301 rows over 2 ranks), config-2 model at real dims
            net, pred, algo, sd, msd = bench.build('cuda:0', 301)
            net.timestep = 9
            t, s = D.sharded_inference(algo, 301, seed=5, score_fn=pred.log_likelihood)
            out = dict(tokens=t.cpu(), scores=s.cpu())
        elif mode == 'cells':       # independent sampler calls dealt to the ranks (config 4 / 5 form), 5 cells of 10 sequences
            net, pred, algo, sd, msd = bench.build('cuda:0', 10)
            net.timestep = 9

            def run_cell(cell, key):
                net.reseed(seed=key, row_offset=0)
                x = algo.inference(num_samples=cell)
                return x, pred.log_likelihood(x)
            res = D.run_cells([10, 7, 10, 3, 10], run_cell, seed=11)
            out = dict(tokens=[t.cpu() for t, _ in res], scores=[s.cpu() for _, s in res])
        elif mode == 'thompson':    # config 5: Thompson draws, two stacked per rank and wave, consumed in draw order
            import numpy as np
            out = thompson_job(D, bench, rank, world, per_rank=2)
        if rank == 0:
            torch.save(out, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
