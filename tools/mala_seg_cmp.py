"""per-segment errors of the H = 64 MALA fixture for a library build (tools only): CCSP_SO=... python tools/mala_seg_cmp.py"""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
from diffusion_ccsp_amd import _lib
if os.environ.get('CCSP_SO'):
    _lib.SO = os.environ['CCSP_SO']; _lib._stale = lambda *a: False
from conftest import golden
import test_hip_parity as T
from test_oracle_golden import MALA_SEGMENTS
z = golden('chain_t64_mala')
den, gd, b = T._golden_chain_model(torch.device('cuda:0'), z)
idx = list(z['hist_idx'])
for i0, i1 in MALA_SEGMENTS:
    k0, k1 = idx.index(i0), idx.index(i1)
    x = gd.p_sample_segment(b, torch.from_numpy(z['hist'][k0]), 999 - i0, 1000 - i1, seed=int(z['seed'])).cpu().numpy()
    want = z['hist'][k1]
    err = np.abs(x - want).max(axis=1)
    print('segment %4d..%4d  rows off (>1e-4) %3d  max err of the other rows %.2e  median err %.2e' % (i0, i1, int((err > 1e-4).sum()), err[err <= 1e-4].max() if (err <= 1e-4).any() else -1, np.median(err)))
