"""s_memtime phase tables of the three evaluation kernels on a small batch (the build with the stamps comes from
tools/trace_build.py).  usage (GPU box): python tools/trace_run.py [c5 | <number of 8-object qualitative graphs>]"""
import os, sys, ctypes as C
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import numpy as np, torch
import diffusion_ccsp_amd
from diffusion_ccsp_amd import _lib, ConstraintDiffuser, GaussianDiffusion, worlds
_lib.SO = os.environ.get('CCSP_SO') or os.path.join(ROOT, 'tools', 'abl_trace.so'); _lib._stale = lambda *a: False
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
buf = np.zeros(3 * 256 * 32, dtype=np.uint64)
L = _lib.lib()
L.ccsp_debug_trace.argtypes = [C.c_void_p]
assert L.ccsp_debug_trace(buf.ctypes.data) == 0
t = buf.reshape(3, 256, 32).astype(np.int64)
names = {0: ['entry', 'index setup'] + ['chunk %d landed' % c for c in range(8)] + ['K loop done (ring)', '-', 'epilogue returned', 'row tile 0 in LDS',
             'row tile 0 stores issued', 'row tile 1 in LDS', 'row tile 1 stores issued', 'stores drained'] +
            ['tile %d group %d computed' % (i, st) for i in range(2) for st in range(4)] + ['tile 0 bt formed', 'tile 1 bt formed'],
         1: ['entry', 'indices+umax', 'stage 0 built'] + ['chunk %d done' % c for c in range(8)] + ['S1 written', 'layer-2 partials', 'O stored'],
         2: ['entry', 'CSR sum + update', 'layer 1', 'layer 2 MFMA', 'row max', 'planes stored', 'update computed (in front of the barrier)', 'node_ptr here', 'noise drawn', 'CSR sum done']}
for kern, title in ((0, 'k_rowgemm_h2'), (1, 'k_edge_h2'), (2, 'k_node')):
    tk = t[kern]
    tk = tk[(tk[:, 0] > 0) & ((tk[:, :30] > 0).sum(axis=1) >= 4)]          # (the row GEMM's noise-ahead workgroups stamp their entry only)
    if not len(tk):
        continue
    nn = names[kern]
    idx = [i for i, n in enumerate(nn) if n != '-' and (tk[:, i] > 0).all()]
    d = tk[:, idx] - tk[:, :1]
    med = np.median(d, axis=0)
    order = np.argsort(med, kind='stable')
    print('%s: %d traced workgroups; cycles since entry (median, p10, p90) and delta of the medians' % (title, len(tk)))
    prev = 0.0
    for j in order:
        print('  %-26s %8.0f %8.0f %8.0f  +%6.0f' % (nn[idx[j]], med[j], np.percentile(d[:, j], 10), np.percentile(d[:, j], 90), med[j] - prev))
        prev = med[j]
    rt = t[kern][t[kern][:, 30] > 0][:, 30:32]
    rt = rt[rt[:, 0] > rt[:, 0].max() - 2000]                   # the last launch (100 MHz ticks: 20 us window)
    e0 = (rt[:, 0] - rt[:, 0].min()) * 10.0
    x1 = (rt[:, 1] - rt[:, 0].min()) * 10.0
    print('  chip-wide clock, ns since the first traced entry (%d workgroups of the last launch): entries median %d p90 %d max %d; exits median %d p90 %d max %d' %
          (len(rt), np.median(e0), np.percentile(e0, 90), e0.max(), np.median(x1), np.percentile(x1, 90), x1.max()))
if os.environ.get('TRACE_RAW'):
    for kern in (0, 1, 2):
        tk = t[kern]; tk = tk[tk[:, 0] > 0]
        e = np.sort(tk[:, 0]); print('raw entry offsets kernel', kern, (e - e.min())[:60].tolist())
