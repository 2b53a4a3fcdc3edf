"""solved fraction of the fixture weights by number of objects (checker.py), 256 graphs each"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import load_weights
from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion, checker, worlds

dev = torch.device('cuda:0')
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
wfile = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'tests', 'golden', 'weights_qualitative_h%d.npz' % H)
den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=H, input_mode='qualitative', EBM='ULA', device=dev, verbose=False)
den.load_state_dict(load_weights(wfile))
gd = GaussianDiffusion(den, timesteps=1000, EBM='ULA', samples_per_step=10)
for n in [int(v) for v in os.environ.get('SWEEP_OBJECTS', '2,3,4,5,6,8').split(',')]:
    b = worlds.qualitative_batch(256, n, seed=11 + n)
    x = gd.sample(b.to_torch(dev), seed=3)
    ok = checker.solved_mask(x.cpu().numpy(), b)
    nan_graphs = len(set(np.asarray(b.batch)[np.isnan(x.cpu().numpy()).any(axis=1)].tolist()))
    print('objects %d: solved %d / %d  (graphs with NaN poses: %d)' % (n, int(ok.sum()), ok.size, nan_graphs), flush=True)
