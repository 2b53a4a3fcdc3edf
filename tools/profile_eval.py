"""profiling helper (not part of the product path): repeats single network evaluations of a BASELINE configuration's
per-GPU batch so that rocprofv3 --pmc passes see the hot kernels in isolation (energy-mode configurations: gradient
evaluations, i.e. all six energy kernels).
usage: python tools/profile_eval.py [n_evals] [graphs] [c2|c4|c5]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault('CCSP_LANES', '1')          # counters of the one-lane, full-batch launches (the variants bench.py's timed pass runs)
from bench import CONFIGS, load_weights
from diffusion_ccsp_amd import ConstraintDiffuser, worlds, _lib
if os.environ.get('CCSP_SO'):          # ablation builds (tools only)
    _lib.SO = os.environ['CCSP_SO']
    _lib._stale = lambda *a: False

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = CONFIGS[sys.argv[3] if len(sys.argv) > 3 else 'c2']
B = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else cfg['graphs']
dev = torch.device('cuda:0')
dims = worlds.MODE_DIMS[cfg['mode']]
den = ConstraintDiffuser(dims=dims, hidden_dim=256, input_mode=cfg['mode'], EBM=cfg['EBM'], energy_wrapper=cfg['energy'], device=dev, verbose=False)
wrel = next(w for w in reversed(cfg['weights']) if os.path.isfile(os.path.join(ROOT, w)))
den.load_state_dict(load_weights(os.path.join(ROOT, wrel)))
batch = getattr(worlds, cfg['batch'])(B, cfg['n_objects'], seed=5).to_torch(dev)
x = (torch.randn(batch.x.shape[0], dims[-1][0]) * 0.7).to(dev)
if cfg['energy']:
    for i in range(n):
        out = den(x, batch, torch.tensor([500 - i]), eval=True, tag='EBM')
        out = out[0] if isinstance(out, tuple) else out
else:
    # direct mode: a few timesteps of the chain itself on ONE lane (the kernels a chain launches, k_node_direct included;
    # n evaluations rounded up to whole timesteps of 1 + S)
    from diffusion_ccsp_amd import GaussianDiffusion
    gd = GaussianDiffusion(den, timesteps=1000, EBM=cfg['EBM'], samples_per_step=10)
    nt = max(1, (n + 10) // 11)
    out = gd.p_sample_segment(batch, x, 500, 500 - nt + 1, seed=3)
torch.cuda.synchronize()
print('ok', float(out.abs().max()))
