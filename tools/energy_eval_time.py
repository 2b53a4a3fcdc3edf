"""A/B helper (tools only): time of one energy-mode gradient evaluation on a qualitative batch (13 constraint types, U rows ~ edges: the
shape where partial rows do not reduce anything) and on the triangular C4 batch.  usage: python tools/energy_eval_time.py [graphs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_weights
from diffusion_ccsp_amd import ConstraintDiffuser, worlds

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device('cuda:0')
for mode, wfile, maker, nobj in (('qualitative', 'weights_qualitative_h256.npz', worlds.qualitative_batch, 8),
                                 ('diffuse_pairwise', 'weights_diffuse_pairwise_h256_energy.npz', worlds.triangular_batch, 12)):
    dims = worlds.MODE_DIMS[mode]
    den = ConstraintDiffuser(dims=dims, hidden_dim=256, input_mode=mode, EBM='MALA', energy_wrapper=True, device=dev, verbose=False)
    den.load_state_dict(load_weights(os.path.join(ROOT, 'tests', 'golden', wfile)))
    batch = maker(B, nobj, seed=5).to_torch(dev)
    x = (torch.randn(batch.x.shape[0], dims[-1][0]) * 0.7).to(dev)
    for i in range(5):
        den(x, batch, torch.tensor([500 - i]), eval=True, tag='EBM')
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        den(x, batch, torch.tensor([500 - i % 50]), eval=True, tag='EBM')
    torch.cuda.synchronize()
    print('%s %d graphs: %.1f us per gradient evaluation (host-paced single calls)' % (mode, B, 1e6 * (time.perf_counter() - t0) / n))
