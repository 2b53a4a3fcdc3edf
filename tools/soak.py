"""race / nondeterminism soak: the C2 chain N times with one seed must be bitwise identical every time (lanes, bf16x3 kernels)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_weights
from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion, worlds

dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=256, input_mode='qualitative', EBM='ULA', device=dev, verbose=False)
den.load_state_dict(load_weights(os.path.join(ROOT, 'tests', 'golden', 'weights_qualitative_h256.npz')))
gd = GaussianDiffusion(den, timesteps=1000, EBM='ULA', samples_per_step=10)
base = worlds.qualitative_batch(256, 8, seed=5).to_torch(dev)
ref = None
for i in range(n):
    x = gd.sample(base.clone(), seed=42)
    if ref is None:
        ref = x.clone()
    same = torch.equal(x, ref)
    print('run %d identical %s finite %s' % (i, same, bool(torch.isfinite(x).all())), flush=True)
    assert same
print('soak ok')
