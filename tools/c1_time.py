"""C1 of BASELINE.json (RandomSplitQualitativeWorld, 3 objects, T=100, batch 1) on the GPU: wall time of a whole chain.
usage (GPU box): python tools/c1_time.py [n_graphs] [n_objects] [T]"""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import diffusion_ccsp_amd
from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion, worlds
from bench import load_weights

n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_obj = int(sys.argv[2]) if len(sys.argv) > 2 else 3
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device('cuda:0')
den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=256, input_mode='qualitative', device=dev, verbose=False)
den.load_state_dict(load_weights(os.path.join(ROOT, 'tests', 'golden', 'weights_qualitative_h256.npz')))
gd = GaussianDiffusion(den, timesteps=T, EBM='ULA', samples_per_step=10)
b = worlds.qualitative_batch(n_graphs, n_obj, seed=5).to_torch(dev)
for _ in range(3):
    gd.sample(b, seed=1)
torch.cuda.synchronize()
ts = []
for i in range(10):
    t0 = time.perf_counter()
    gd.sample(b, seed=2 + i)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
st = gd.chain_stats()
print('graphs %d objects %d T %d: chain %.2f ms (min of 10; median %.2f), %d evaluations, %.1f us per evaluation' %
      (n_graphs, n_obj, T, 1e3 * min(ts), 1e3 * sorted(ts)[5], st['evals'], 1e6 * min(ts) / max(st['evals'], 1)))
