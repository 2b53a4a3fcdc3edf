import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_weights
from diffusion_ccsp_amd import ConstraintDiffuser, worlds
dev = torch.device('cuda:0')
den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=256, input_mode='qualitative', device=dev, verbose=False)
den.load_state_dict(load_weights(os.path.join(ROOT, 'tests', 'golden', 'weights_qualitative_h256.npz')))
base = worlds.qualitative_batch(256, 8, seed=5).to_torch(dev)
for i in range(5):
    b = base.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g = den._graph(b)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    den._graphs.clear(); del g
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print('graph_create %.2f ms   destroy %.2f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
