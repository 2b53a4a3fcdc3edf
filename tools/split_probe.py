"""experiment: run the chain of a 256-graph batch as K concurrent sub-batch chains on K streams"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import load_weights
from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion, worlds, sharding

dev = torch.device('cuda:0')
den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=256, input_mode='qualitative', EBM='ULA', device=dev, verbose=False)
den.load_state_dict(load_weights(os.path.join(ROOT, 'tests', 'golden', 'weights_qualitative_h256.npz')))
gd = GaussianDiffusion(den, timesteps=1000, EBM='ULA', samples_per_step=10)
full = worlds.qualitative_batch(256, 8, seed=5)
ref = None
for K in (1, 2, 4):
    subs = [sharding.shard_batch(full, r, K) for r in range(K)]
    tb = [(s.to_torch(dev), off) for s, off in subs]
    streams = [torch.cuda.Stream(dev) for _ in range(K)]
    for b, _ in tb:
        den._graph(b)
    torch.cuda.synchronize()
    outs = [None] * K
    def work(i):
        with torch.cuda.stream(streams[i]):
            outs[i] = gd.p_sample_loop(tb[i][0], seed=42, row_offset=tb[i][1])
    for rep in range(2):
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(K)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    x = torch.cat(outs).cpu().numpy()
    if ref is None: ref = x
    print('K=%d  %.1f ms  %.1f samples/s   max diff vs K=1: %.2e' % (K, dt * 1e3, 256 / dt, np.abs(x - ref).max()))
