"""profiling helper: single evaluations of the StructDiffusion baseline (256 graphs x 8 tokens, H=256) for rocprofv3"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diffusion_ccsp_amd import ConstraintDiffuser, worlds

dev = torch.device('cuda:0')
den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=256, input_mode='qualitative', device=dev, verbose=False,
                         model='StructDiffusion')
den.reset_parameters(0)
b = worlds.qualitative_batch(256, 7, seed=4).to_torch(dev)
x = (torch.randn(b.x.shape[0], 4) * 0.7).to(dev)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    out = den(x, b, torch.tensor([500 - i]), eval=True)
torch.cuda.synchronize()
print('ok', float(out.abs().max()))
