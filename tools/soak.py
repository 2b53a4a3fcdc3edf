"""race / nondeterminism soak: the chain of a bench configuration N times with one seed must be bitwise identical every time
(c2: lanes + the throughput forms of the kernels; c5 / small: the ring row GEMM with counted waits and bare barriers, the split
decoder layer; c4: energy mode, MALA).   usage (GPU box): python tools/soak.py [n] [c2|c4|c5|small]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import CONFIGS, load_weights
from diffusion_ccsp_amd import ComposedEBMDenoiseFn, ConstraintDiffuser, GaussianDiffusion, worlds

dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
name = sys.argv[2] if len(sys.argv) > 2 else 'c2'
cfg = dict(CONFIGS['c2' if name == 'small' else name])
if name == 'small':
    cfg.update(graphs=12, n_objects=5)
den = ConstraintDiffuser(dims=worlds.MODE_DIMS[cfg['mode']], hidden_dim=256, input_mode=cfg['mode'], EBM=cfg['EBM'],
                         energy_wrapper=cfg['energy'], device=dev, verbose=False)
wfile = cfg['weights'][-1] if name != 'c2' and name != 'small' else 'tests/golden/weights_qualitative_h256.npz'
den.load_state_dict(load_weights(os.path.join(ROOT, wfile)))
gd = GaussianDiffusion(ComposedEBMDenoiseFn(den) if cfg['energy'] else den, timesteps=1000, EBM=cfg['EBM'], samples_per_step=10)
base = getattr(worlds, cfg['batch'])(cfg['graphs'], cfg['n_objects'], seed=5).to_torch(dev)
ref = None
for i in range(n):
    x = gd.sample(base.clone(), seed=42)
    if ref is None:
        ref = x.clone()
    same = torch.equal(x, ref) or bool(((x == ref) | (torch.isnan(x) & torch.isnan(ref))).all())
    print('%s run %d identical %s finite %s' % (name, i, same, bool(torch.isfinite(x).all())), flush=True)
    assert same
print('soak ok:', name)
