import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from sgpo_amd import runtime
runtime.init('cuda:0')
import test_gpu_graph as T
net, pred, algo = T._sampler('cg_real_guided', 0.064)
B = 10
def run(graphs, seed, row_offset=0):
    net.use_graphs = graphs
    net.reseed(seed, row_offset=row_offset)
    return algo.inference(num_samples=B)
for ro in (0, 40, 4, 1):
    e = run(False, 12, ro); e2 = run(False, 12, ro); g = run(True, 12, ro); g2 = run(True, 12, ro)
    print('row_offset', ro, 'eager==eager', torch.equal(e, e2), 'graph==graph', torch.equal(g, g2), 'eager==graph', torch.equal(e, g),
          'ndiff', int((e != g).sum()), 'rows differing', (e != g).any(1).nonzero().flatten().tolist())
# step-level: first differing step
net.use_graphs = False
