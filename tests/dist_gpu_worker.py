"""Rank body of tests/test_gpu_dist.py (not collected by pytest): one process of a world-2 job on ONE GPU over gloo.
usage: dist_gpu_worker.py <mode> <out.pt>   with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


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
        if rank == 0:
            torch.save(out, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
