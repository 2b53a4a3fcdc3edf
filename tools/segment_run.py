"""a short segment of a C2-shaped chain (tools only): python tools/segment_run.py <graphs> [timesteps]"""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import torch
from diffusion_ccsp_amd import _lib, ConstraintDiffuser, GaussianDiffusion, worlds
if os.environ.get('CCSP_SO'):
    _lib.SO = os.environ['CCSP_SO']; _lib._stale = lambda *a: False
from bench import load_weights
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 6
den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=256, input_mode='qualitative', device=dev, verbose=False)
den.load_state_dict(load_weights(os.path.join(ROOT, 'tests', 'golden', 'weights_qualitative_h256.npz')))
gd = GaussianDiffusion(den, timesteps=1000, EBM='ULA', samples_per_step=10)
b = worlds.qualitative_batch(n, 8, seed=5).to_torch(dev)
x0 = torch.zeros(b.x.shape[0], 4, device=dev)
for rep in range(2):
    x = gd.p_sample_segment(b, x0, 500, 500 - nt + 1, seed=3)
torch.cuda.synchronize()
print('ok', float(x.abs().max()))
