import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_weights
from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion, worlds
dev = torch.device('cuda:0')
den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=256, input_mode='qualitative', EBM='ULA', device=dev, verbose=False)
den.load_state_dict(load_weights(os.path.join(ROOT, 'tests', 'golden', 'weights_qualitative_h256.npz')))
gd = GaussianDiffusion(den, timesteps=1000, EBM='ULA', samples_per_step=10)
b = worlds.qualitative_batch(256, 8, seed=5).to_torch(dev)
for rep in range(3):
    t0 = time.perf_counter(); gd.sample(b, seed=1); dt = time.perf_counter() - t0
    print('lanes=%s reuse graph: %.1f ms' % (os.environ.get('CCSP_LANES', 'default'), dt * 1e3))
for rep in range(2):
    t0 = time.perf_counter(); gd.sample(b.clone(), seed=1); dt = time.perf_counter() - t0
    print('lanes=%s fresh graph: %.1f ms' % (os.environ.get('CCSP_LANES', 'default'), dt * 1e3))
