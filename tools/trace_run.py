"""s_memtime phase tables of the three evaluation kernels on a small batch (the build with the stamps comes from
tools/trace_build.py).  usage (GPU box): python tools/trace_run.py [c5 | <number of 8-object qualitative graphs>]"""
import os, sys, ctypes as C
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import numpy as np, torch
import diffusion_ccsp_amd
from diffusion_ccsp_amd import _lib, ConstraintDiffuser, GaussianDiffusion, worlds
_lib.SO = os.path.join(ROOT, 'tools', 'abl_trace.so'); _lib._stale = lambda: False
from bench import load_weights
dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else 'c5'
if which == 'c5':
    mode, wf, batch = 'robot_box', 'weights_robot_box_h256.npz', worlds.robot_box_batch(64, 10, seed=5)
else:
    mode, wf, batch = 'qualitative', 'weights_qualitative_h256.npz', worlds.qualitative_batch(int(which), 8, seed=5)
den = ConstraintDiffuser(dims=worlds.MODE_DIMS[mode], hidden_dim=256, input_mode=mode, device=dev, verbose=False)
den.load_state_dict(load_weights(os.path.join(ROOT, 'tests', 'golden', wf)))
gd = GaussianDiffusion(den, timesteps=1000, EBM='ULA', samples_per_step=10)
b = batch.to_torch(dev)
x0 = torch.zeros(b.x.shape[0], worlds.MODE_DIMS[mode][1][0], device=dev)
x = gd.p_sample_segment(b, x0, 500, 495, seed=3)
torch.cuda.synchronize()
buf = np.zeros(3 * 64 * 32, dtype=np.uint64)
L = _lib.lib()
L.ccsp_debug_trace.argtypes = [C.c_void_p]
assert L.ccsp_debug_trace(buf.ctypes.data) == 0
t = buf.reshape(3, 64, 32).astype(np.int64)
names = {0: ['entry', 'index setup'] + ['chunk %d landed' % c for c in range(8)] + ['K loop done', '-', 'epilogue done'],
         1: ['entry', 'indices+umax', 'stage 0 built'] + ['chunk %d done' % c for c in range(8)] + ['S1 written', 'layer-2 partials', 'O stored'],
         2: ['entry', 'CSR sum + update', 'layer 1', 'layer 2 MFMA', 'row max', 'planes stored']}
for kern, title in ((0, 'k_rowgemm_h2 ring'), (1, 'k_edge_h2'), (2, 'k_node')):
    tk = t[kern]
    live = tk[:, 0] > 0
    tk = tk[live]
    nn = names[kern]
    idx = [i for i, n in enumerate(nn) if n != '-']
    d = tk[:, idx] - tk[:, :1]
    med = np.median(d, axis=0)
    print('%s: %d traced workgroups; cycles since entry (median) and delta' % (title, len(tk)))
    for j, i in enumerate(idx):
        print('  %-18s %8.0f  +%6.0f' % (nn[i], med[j], med[j] - (med[j - 1] if j else 0)))
    print('  span of entry times %d, of exit times %d' % (tk[:, 0].max() - tk[:, 0].min(), tk[:, idx[-1]].max() - tk[:, idx[-1]].min()))
