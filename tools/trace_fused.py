"""s_memtime phase table of k_eval_fused (tools/trace_build.py builds the library with the stamps).
usage (GPU box): CCSP_EVAL=fused python tools/trace_fused.py [number of 8-object qualitative graphs | c5]"""
import os, sys, ctypes as C
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
os.environ['CCSP_EVAL'] = 'fused'
os.environ.setdefault('CCSP_LANES', '1')
import numpy as np, torch
import diffusion_ccsp_amd
from diffusion_ccsp_amd import _lib, ConstraintDiffuser, GaussianDiffusion, worlds
_lib.SO = os.environ.get('CCSP_SO') or os.path.join(ROOT, 'tools', 'abl_trace.so'); _lib._stale = lambda *a: False
from bench import load_weights
dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else '256'
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
nn = ['entry', 'index loads used', 'A planes landed (barrier)', 'k-steps 0-3', 'k-steps 4-7', 'k-steps 8-11', 'k-steps 12-15', 'base landed (barrier)',
      'U tile written (barrier)', 'row maxima (barrier)', 'stage 0 built (barrier)', '-', '-', '-', 'decoder K loop done', 'S1 written (barrier)', 'O stored']
tk = t[0]
tk = tk[tk[:, 0] > 0]
idx = [i for i, n in enumerate(nn) if n != '-' and (tk[:, i] > 0).all()]
d = tk[:, idx] - tk[:, :1]
med = np.median(d, axis=0)
print('k_eval_fused (%s): %d traced workgroups (blockIdx %% 8 == 0); cycles since entry (median, p10, p90), delta of the medians' % (which, len(tk)))
prev = 0.0
for j in range(len(idx)):
    print('  %-30s %8.0f %8.0f %8.0f  +%6.0f' % (nn[idx[j]], med[j], np.percentile(d[:, j], 10), np.percentile(d[:, j], 90), med[j] - prev))
    prev = med[j]
life = tk[:, 16] - tk[:, 0]
print('  lifetime: median %d, p10 %d, p90 %d, max %d' % (np.median(life), np.percentile(life, 10), np.percentile(life, 90), life.max()))
rt = t[0][t[0][:, 30] > 0][:, 30:32]
rt = rt[rt[:, 0] > rt[:, 0].max() - 10000]
e0 = (rt[:, 0] - rt[:, 0].min()) * 10.0
x1 = (rt[:, 1] - rt[:, 0].min()) * 10.0
print('  chip-wide clock, ns since the first traced entry (%d workgroups of the last launch): entries median %d p90 %d max %d; exits median %d p90 %d max %d' %
      (len(rt), np.median(e0), np.percentile(e0, 90), e0.max(), np.median(x1), np.percentile(x1, 90), x1.max()))
