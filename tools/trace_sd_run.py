"""s_memtime phase tables of the transformer baseline's GEMM k_sd_gemm_h2 (build: tools/trace_build.py, -DCCSP_TRACE): where a workgroup's life goes --
entry -> first operands landed -> first stage built -> K loop done -> prefetch drained -> accumulators in LDS -> stores / row maxima done.
Slots: 0 = the last BIAS-epilogue launch of the evaluation (c_proj of block 3, split K), 1 = c_fc (QuickGELU, 64 x 128 tiles), 2 = out_proj (residual).
usage (GPU box): python tools/trace_sd_run.py [graphs=256] [lanes=1]"""
import os, sys, ctypes as C
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
os.environ['CCSP_LANES'] = sys.argv[2] if len(sys.argv) > 2 else '1'
import numpy as np, torch
import diffusion_ccsp_amd
from diffusion_ccsp_amd import _lib, ConstraintDiffuser, worlds
_lib.SO = os.environ.get('CCSP_SO') or os.path.join(ROOT, 'tools', 'abl_trace.so'); _lib._stale = lambda *a: False
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=256, input_mode='qualitative', device=dev, verbose=False, model='StructDiffusion')
den.reset_parameters(0)
b = worlds.qualitative_batch(B, 7, seed=4).to_torch(dev)
x = (torch.randn(b.x.shape[0], 4) * 0.7).to(dev)
for i in range(3):
    out = den(x, b, torch.tensor([500 - i]), eval=True)
torch.cuda.synchronize()
buf = np.zeros(3 * 256 * 32, dtype=np.uint64)
L = _lib.lib()
L.ccsp_debug_trace.argtypes = [C.c_void_p]
assert L.ccsp_debug_trace(buf.ctypes.data) == 0
t = buf.reshape(3, 256, 32).astype(np.int64)
names = ['entry', 'first operands + row maxima landed', 'stage 0 built, first barrier', 'K loop done', 'prefetch drained', 'accumulators in LDS', 'stores + row maxima issued']
for kern, title in ((0, 'BIAS epilogue (last launch: c_proj of block 3)'), (1, 'c_fc (QuickGELU, 64 x 128 tiles)'), (2, 'out_proj (residual)')):
    tk = t[kern]
    tk = tk[(tk[:, 0] > 0) & (tk[:, 6] > 0)]
    if not len(tk):
        continue
    d = tk[:, :7] - tk[:, :1]
    med = np.median(d, axis=0)
    print('%s: %d traced workgroups; cycles since entry (median, p10, p90) and delta of the medians' % (title, len(tk)))
    prev = 0.0
    for j in range(7):
        print('  %-38s %8.0f %8.0f %8.0f  +%6.0f' % (names[j], med[j], np.percentile(d[:, j], 10), np.percentile(d[:, j], 90), med[j] - prev))
        prev = med[j]
    rt = t[kern][t[kern][:, 30] > 0][:, 30:32]
    rt = rt[rt[:, 0] > rt[:, 0].max() - 4000]
    e0 = (rt[:, 0] - rt[:, 0].min()) * 10.0
    x1 = (rt[:, 1] - rt[:, 0].min()) * 10.0
    print('  chip-wide clock, ns since the first traced entry (%d workgroups of the last launch): entries median %d p90 %d max %d; exits median %d p90 %d max %d' %
          (len(rt), np.median(e0), np.percentile(e0, 90), e0.max(), np.median(x1), np.percentile(x1, 90), x1.max()))
