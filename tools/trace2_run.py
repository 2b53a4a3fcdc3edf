"""Phase timelines of the workgroups that SHARE a compute unit in k_rowgemm_h2 (build: tools/trace2_build.py, -DCCSP_TRACE2).
usage (GPU box): python tools/trace2_run.py [graphs=256] [lanes=1]        (C2 batch: 8-object qualitative graphs)
Per workgroup: s_memtime (low 32 bits) at entry, at the begin / end of each K chunk's MFMA phase, at the end of the K loop, after the epilogue's
last store was issued and after the stores were acknowledged; HW_ID / XCC_ID of the wave that stamped last.  Workgroups are grouped by
(XCC, SE, SH, CU); s_memtime is one counter per shader engine, so the stamps of co-resident workgroups are on one clock."""
import os, sys, ctypes as C
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
lanes = sys.argv[2] if len(sys.argv) > 2 else '1'
os.environ['CCSP_LANES'] = lanes
import numpy as np, torch
import diffusion_ccsp_amd
from diffusion_ccsp_amd import _lib, ConstraintDiffuser, GaussianDiffusion, worlds
_lib.SO = os.environ.get('CCSP_SO') or os.path.join(ROOT, 'tools', 'abl_trace2.so'); _lib._stale = lambda *a: False
from bench import load_weights
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
batch = worlds.qualitative_batch(B, 8, seed=5)
den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=256, input_mode='qualitative', device=dev, verbose=False)
den.load_state_dict(load_weights(os.path.join(ROOT, 'tests', 'golden', 'weights_qualitative_h256.npz')))
gd = GaussianDiffusion(den, timesteps=1000, EBM='ULA', samples_per_step=10)
b = batch.to_torch(dev)
x0 = torch.zeros(b.x.shape[0], 4, device=dev)
x = gd.p_sample_segment(b, x0, 900, 500 + int(os.environ.get('TRACE_WARM', '0')), seed=3) if os.environ.get('TRACE_WARM') else x0      # (clocks ramp with load)
x = gd.p_sample_segment(b, x0, 500, 496, seed=3)
torch.cuda.synchronize()
print('finite outputs:', bool(torch.isfinite(x).all()), ' row GEMM mode / edge tile:', gd.kernel_variant())
buf = np.zeros(4096 * 40, dtype=np.uint32)
L = _lib.lib()
L.ccsp_debug_trace2.argtypes = [C.c_void_p]
assert L.ccsp_debug_trace2(buf.ctypes.data) == 0
t = buf.reshape(4096, 40).astype(np.int64)
live = t[:, 0] != 0
t = t[live]
print('%d traced workgroups (last launch of k_rowgemm_h2 on every block index)' % len(t))
hw, xcc = t[:, 36], t[:, 37] & 0xf
cu, sh, se, simd = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 4) & 3
key = xcc * 1000 + se * 100 + sh * 16 + cu
NCH = 8
def rel(a, b):          # a - b on the 32-bit counter
    return ((a - b + (1 << 31)) % (1 << 32)) - (1 << 31)
groups = {}
for i, k in enumerate(key):
    groups.setdefault(int(k), []).append(i)
sizes = np.bincount([len(v) for v in groups.values()])
print('compute units seen: %d; workgroups per CU histogram (index = count): %s' % (len(groups), sizes.tolist()))
# per-workgroup durations
dur = rel(t[:, 27], t[:, 0])
kloop = rel(t[:, 25], t[:, 1])
pro = rel(t[:, 1], t[:, 0])
epi = rel(t[:, 26], t[:, 25])
drain = rel(t[:, 27], t[:, 26])
comp = np.stack([rel(t[:, 2 + 2 * c], t[:, 1 + 2 * c]) for c in range(NCH)], 1)          # MFMA phase of chunk c
gap = np.stack([rel(t[:, 3 + 2 * c], t[:, 2 + 2 * c]) for c in range(NCH - 1)], 1)       # barrier + ds_write + loads + barrier between chunks
q = lambda v: '%6.0f / %6.0f / %6.0f' % (np.percentile(v, 10), np.median(v), np.percentile(v, 90))
print('cycles per workgroup (p10 / median / p90):')
print('  whole workgroup            %s' % q(dur))
print('  prologue (entry -> chunk 0) %s' % q(pro))
print('  K loop                      %s' % q(kloop))
print('    MFMA phase of a chunk     %s   (12 MFMA 32x32x16 x 2 k-steps per wave = 768 pipe cycles; x waves per SIMD when they coincide)' % q(comp.ravel()))
print('    between chunks            %s   (barrier, 8 ds_write_b128 per thread, next loads issued, barrier)' % q(gap.ravel()))
print('  epilogue (issue)            %s' % q(epi))
print('  stores acknowledged         %s' % q(drain))
# co-residency: how the phases of the workgroups on ONE compute unit lie against each other
frac_any, frac_mean, span_all, phase_spread = [], [], [], []
for k, idx in groups.items():
    if len(idx) < 2:
        continue
    base = t[idx[0], 0]
    ent = np.array([rel(t[i, 0], base) for i in idx])
    t0 = ent.min()
    end = max(rel(t[i, 27], base) for i in idx)
    ev = []
    for i in idx:
        for c in range(NCH):
            ev.append((rel(t[i, 1 + 2 * c], base), +1))
            ev.append((rel(t[i, 2 + 2 * c], base), -1))
    ev.sort()
    busy, wsum, n, last = 0, 0, 0, t0
    for tt, d in ev:
        if n > 0:
            busy += tt - last
        wsum += n * (tt - last)
        n += d
        last = tt
    span = end - t0
    frac_any.append(busy / span); frac_mean.append(wsum / span); span_all.append(span)
    # phase offset of the workgroups' chunk-3 start relative to the CU's mean chunk period
    per = np.median([rel(t[i, 1 + 2 * 4], t[i, 1 + 2 * 3]) for i in idx])
    st = np.array([rel(t[i, 1 + 2 * 3], base) for i in idx], dtype=np.float64)
    ph = ((st - st.min()) % per) / per
    phase_spread.append(ph.max())
print('co-resident workgroups (CUs with >= 2): %d CUs' % len(frac_any))
print('  CU busy span (first entry -> last store acknowledged), cycles   %s' % q(np.array(span_all)))
print('  fraction of that span with >= 1 workgroup in an MFMA phase       %s' % q(100 * np.array(frac_any)))
print('  mean number of workgroups in an MFMA phase at once               %.2f' % float(np.mean(frac_mean)))
print('  largest phase offset between co-residents at chunk 3 (fraction of a chunk period; 0 = locked, ~0.67 = evenly spread over 3)  p10 %.2f median %.2f p90 %.2f'
      % (np.percentile(phase_spread, 10), np.median(phase_spread), np.percentile(phase_spread, 90)))
# a few CUs in full
shown = 0
for k, idx in sorted(groups.items()):
    if len(idx) < 3 or shown >= 4:
        continue
    shown += 1
    base = min(t[i, 0] for i in idx)
    print('CU xcc %d se %d sh %d cu %d: %d workgroups' % (k // 1000, (k // 100) % 10, (k % 100) // 16, k % 16, len(idx)))
    for i in sorted(idx, key=lambda i: rel(t[i, 0], base)):
        row = [rel(t[i, 0], base)] + [rel(t[i, 1 + 2 * c], base) for c in range(NCH)] + [rel(t[i, 25], base), rel(t[i, 26], base), rel(t[i, 27], base)]
        ends = [rel(t[i, 2 + 2 * c], base) for c in range(NCH)]
        print('   entry %6d | chunk begins %s | chunk ends %s | K done %6d  epilogue issued %6d  drained %6d' %
              (row[0], ' '.join('%6d' % v for v in row[1:9]), ' '.join('%6d' % v for v in ends), row[9], row[10], row[11]))
if os.environ.get('TRACE_DUMP'):
    np.save(os.environ['TRACE_DUMP'], t)
