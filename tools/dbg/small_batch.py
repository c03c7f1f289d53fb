"""Latency of one guided inference() (127 reverse steps) at small batches (config 5: per-member batches of 10)."""
import sys, os, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench
for B in (10, 100, 400):
    net, pred, algo, sd, msd = bench.build('cuda:0', B)
    net.reseed(seed=1, row_offset=0)
    algo.inference(num_samples=B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3):
        net.reseed(seed=2 + i, row_offset=0)
        x = algo.inference(num_samples=B)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f'B={B}: {dt * 1e3:.1f} ms per inference ({dt / 127 * 1e6:.0f} us per reverse step), {B / dt:.1f} sequences/s', flush=True)
