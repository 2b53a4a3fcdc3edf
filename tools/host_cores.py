"""what the GPU box's host gives this container: os.cpu_count, the affinity mask, the cgroup quota, and how the oracle's OpenMP evaluation scales with
threads (tools only)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
print('os.cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try:
        print(f, open(f).read().strip())
    except OSError:
        pass
print('loadavg', open('/proc/loadavg').read().strip())
import ctypes, numpy as np
import oracle
from diffusion_ccsp_amd import worlds
W = oracle.load_weights(os.path.join(ROOT, 'tests', 'golden', 'weights_diffuse_pairwise_h256_energy.npz'))
m = oracle.OracleModel(W, worlds.MODE_DIMS['diffuse_pairwise'], 256, 2, timesteps=1000, energy_wrapper=True, samples_per_step=10)
b = worlds.triangular_batch(64, 12, seed=5).to_torch()
g = m.graph(b)
x = (np.random.default_rng(0).standard_normal((b.x.shape[0], 4)) * 0.5).astype(np.float32)
gomp = ctypes.CDLL('libgomp.so.1')
for n in (1, 4, 8, 16, 32, 64, 128, 256):
    gomp.omp_set_num_threads(n)
    g.energy_grad(x, 300)
    t0 = time.time()
    for _ in range(3):
        g.energy_grad(x, 300)
    print('threads %3d: %.3f s per gradient evaluation of 64 x 12 triangles' % (n, (time.time() - t0) / 3))
