"""profiling helper (not part of the product path): repeats single network evaluations of the C2
batch so that rocprofv3 --pmc passes see the hot kernels in isolation.
usage: python tools/profile_eval.py [n_evals] [graphs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import load_weights
from diffusion_ccsp_amd import ConstraintDiffuser, worlds, _lib
if os.environ.get('CCSP_SO'):          # ablation builds (tools only)
    _lib.SO = os.environ['CCSP_SO']
    _lib._stale = lambda: False

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device('cuda:0')
den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=256, input_mode='qualitative', device=dev, verbose=False)
den.load_state_dict(load_weights(os.path.join(ROOT, 'tests', 'golden', 'weights_qualitative_h256.npz')))
batch = worlds.qualitative_batch(B, 8, seed=5).to_torch(dev)
x = (torch.randn(batch.x.shape[0], 4) * 0.7).to(dev)
for i in range(n):
    out = den(x, batch, torch.tensor([500 - i]), eval=True)
torch.cuda.synchronize()
print('ok', float(out.abs().max()))
