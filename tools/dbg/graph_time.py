"""Eager vs recorded-graph wall clock of Classifier_Guidance_Inpaint.inference at small batches (real DiT dims, CreiLOV)."""
import sys, os, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench
net, pred, algo, sd, msd = bench.build('cuda:0', 64)
for temp in (0.0, bench.TEMPERATURE):
    algo.temperature = temp
    for B in (10, 64, 200):
        res = {}
        for graphs in (False, True):
            net.use_graphs = graphs
            net.reseed(1); algo.inference(num_samples=B); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(3):
                net.reseed(2 + i); x = algo.inference(num_samples=B)
            torch.cuda.synchronize()
            res[graphs] = (time.perf_counter() - t0) / 3 * 1e3
        print(f'temperature {temp}: B = {B}: eager {res[False]:.1f} ms, graph {res[True]:.1f} ms per inference ({res[False] / res[True]:.2f}x)')
