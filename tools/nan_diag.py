import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
from bench import load_weights
from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion, worlds
dev = torch.device('cuda:0')
W = load_weights(os.path.join(ROOT, 'weights', 'qualitative_h256_ref30k.npz'))
den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=256, input_mode='qualitative', EBM='ULA', device=dev, verbose=False)
den.load_state_dict(W)
gd = GaussianDiffusion(den, timesteps=1000, EBM='ULA', samples_per_step=10)
b = worlds.qualitative_batch(16, 8, seed=19)
x, hist = gd.sample(b.to_torch(dev), seed=3, return_history=True)
h = torch.stack(hist).cpu().numpy()
bad = np.isnan(h).any(axis=(1, 2))
print('mode', os.environ.get('CCSP_MMA'), 'first NaN at history index', int(np.argmax(bad)) if bad.any() else None, 'final NaN rows', int(np.isnan(h[-1]).any(axis=1).sum()), '/', h.shape[1])
mx = np.nanmax(np.abs(h), axis=(1, 2))
print('max |x| at idx 0,1,2,5,10,50,100,500,900,999,1000:', [float(mx[i]) for i in (0, 1, 2, 5, 10, 50, 100, 500, 900, 999, 1000)])
if bad.any():
    k = int(np.argmax(bad)); rows = np.nonzero(np.isnan(h[k]).any(axis=1))[0]; print('rows', rows[:10], 'prev state of first row', h[k - 1][rows[0]], 'mask', b.mask[rows[0]])
import oracle
om = oracle.OracleModel(W, worlds.MODE_DIMS['qualitative'], 256, 13)
b1 = worlds.qualitative_batch(2, 8, seed=19)
out, hh = om.graph(b1).chain('ULA', seed=3, history=True)
print('oracle (2 graphs): final NaN rows', int(np.isnan(out).any(axis=1).sum()), 'max', float(np.nanmax(np.abs(hh))))
x2 = gd.sample(b1.to_torch(dev), seed=3).cpu().numpy()
print('hip same 2 graphs: NaN rows', int(np.isnan(x2).any(axis=1).sum()), 'max diff vs oracle', float(np.nanmax(np.abs(x2 - out))))
