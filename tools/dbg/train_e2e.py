"""End-to-end time of the training drop-in (host work included): 10 members, 500 sequences of 119 residues,
batch 64, time-conditioned (q_sample through the MDLM drop-in + project_fn)."""
import sys, os, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench
net, pred, algo, sd, msd = bench.build('cuda:0', 16)
from problem.protein import ProteinPredictor
from training.train_classifier import train_classifier_ensemble
dc = dict(alphabet=bench.ALPHABET, alphabet_size=31, seq_len=119, full_seq=bench.CREILOV, residues=None)
g = torch.Generator().manual_seed(0)
n, epochs = 500, int(sys.argv[1]) if len(sys.argv) > 1 else 20
X = torch.randint(0, 20, (n, 119), generator=g)
Y = torch.randn(n, generator=g)
loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=64, shuffle=True)
tc = dict(learning_rate=1e-3, wandb=False, n_epochs=epochs, n_epochs_no_time=epochs, batch_size=64)
clfs = [ProteinPredictor(dc, dict(hidden_dim=64), device='cuda:0') for _ in range(10)]
train_classifier_ensemble(clfs, net, loader, dict(tc, n_epochs=1), project_fn=algo.project)
torch.cuda.synchronize()
t0 = time.perf_counter()
train_classifier_ensemble(clfs, net, loader, tc, project_fn=algo.project)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
steps = epochs * ((n + 63) // 64)
print(f'{steps} steps x 10 members in {dt:.2f} s: {dt / steps * 1e6:.0f} us per step end to end '
      f'(1000 epochs would take {dt / epochs * 1000:.0f} s)')
