"""s_memtime phase table of the fused decoder kernel k_edge_fb_h2 (energy mode, round 6) on the C4 batch (build: tools/trace_build.py, -DCCSP_TRACE).
usage (GPU box): python tools/trace_fb_run.py [graphs=256]"""
import os, sys, ctypes as C
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import numpy as np, torch
import diffusion_ccsp_amd
from diffusion_ccsp_amd import _lib, ConstraintDiffuser, worlds
_lib.SO = os.environ.get('CCSP_SO') or os.path.join(ROOT, 'tools', 'abl_trace.so'); _lib._stale = lambda *a: False
from bench import load_weights
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
den = ConstraintDiffuser(dims=worlds.MODE_DIMS['diffuse_pairwise'], hidden_dim=256, input_mode='diffuse_pairwise', EBM='MALA', energy_wrapper=True, device=dev, verbose=False)
den.load_state_dict(load_weights(os.path.join(ROOT, 'tests', 'golden', 'weights_diffuse_pairwise_h256_energy.npz')))
b = worlds.triangular_batch(B, 12, seed=5).to_torch(dev)
x = (torch.randn(b.x.shape[0], 4) * 0.7).to(dev)
for i in range(4):
    out = den(x, b, torch.tensor([500 - i]), eval=True, tag='EBM')
torch.cuda.synchronize()
buf = np.zeros(3 * 256 * 32, dtype=np.uint64)
L = _lib.lib()
L.ccsp_debug_trace.argtypes = [C.c_void_p]
assert L.ccsp_debug_trace(buf.ctypes.data) == 0
t = buf.reshape(3, 256, 32).astype(np.int64)[1]
names = ['entry', 'indices, umax, first stage built', 'forward K loop done (8 chunks)', 'forward epilogue: S1, layer 2, energy partial, go', 'A planes built (once)',
         'backward pass 0 K loop done (4 chunks)', 'backward pass 1 K loop done', 'epilogue pass 0: x SiLU\'(z), partial rows, planes stored', '-', 'epilogue pass 1 done',
         'pass 0: U rows requested, accumulators in the tile', 'pass 0: x SiLU\'(z) written back', 'pass 0: exponents of the partial rows', '-',
         'pass 1: U rows requested, accumulators in the tile', 'pass 1: x SiLU\'(z) written back', 'pass 1: (exponents: pass 0\'s)']
tk = t[(t[:, 0] > 0) & (t[:, 9] > 0)]
idx = [i for i, n in enumerate(names) if n != '-']
d = tk[:, idx] - tk[:, :1]
med = np.median(d, axis=0)
print('k_edge_fb_h2: %d traced workgroups of %d graphs; cycles since entry (median, p10, p90) and delta of the medians' % (len(tk), B))
prev = 0.0
for j in np.argsort(med, kind='stable'):
    print('  %-62s %8.0f %8.0f %8.0f  +%6.0f' % (names[idx[j]], med[j], np.percentile(d[:, j], 10), np.percentile(d[:, j], 90), med[j] - prev))
    prev = med[j]
rt = t[t[:, 30] > 0][:, 30:32]
rt = rt[rt[:, 0] > rt[:, 0].max() - 6000]
e0 = (rt[:, 0] - rt[:, 0].min()) * 10.0
x1 = (rt[:, 1] - rt[:, 0].min()) * 10.0
print('  chip-wide clock, ns since the first traced entry (%d workgroups of the last launch): entries median %d p90 %d max %d; exits median %d p90 %d max %d' %
      (len(rt), np.median(e0), np.percentile(e0, 90), e0.max(), np.median(x1), np.percentile(x1, 90), x1.max()))
