"""Kernel time against wall time for B = 10 guided sampling (is the small-batch path launch-bound?)."""
import sys, os, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
net, pred, algo, sd, msd = bench.build('cuda:0', B)
net.reseed(seed=1, row_offset=0)
algo.inference(num_samples=B)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3):
    net.reseed(seed=2 + i, row_offset=0)
    algo.inference(num_samples=B)
torch.cuda.synchronize()
print('WALL_MS_PER_INFERENCE', (time.perf_counter() - t0) / 3 * 1e3)
