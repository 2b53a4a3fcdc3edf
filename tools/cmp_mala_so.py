"""A/B helper (tools only): the C4 MALA chain (256 x 12 triangles, T = 1000) on two builds of the library, compared bit for bit
(final poses and acceptance rates).  usage: python tools/cmp_mala_so.py <a.so> <b.so> [graphs]"""
import os, sys, subprocess, numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, ROOT)
if sys.argv[1] == '--run':
    import torch
    from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion, worlds, _lib
    _lib.SO = os.environ['CCSP_SO']; _lib._stale = lambda *a: False
    from bench import load_weights
    dev = torch.device('cuda:0')
    den = ConstraintDiffuser(dims=worlds.MODE_DIMS['diffuse_pairwise'], hidden_dim=256, input_mode='diffuse_pairwise', EBM='MALA', energy_wrapper=True,
                             device=dev, verbose=False)
    den.load_state_dict(load_weights(os.path.join(ROOT, 'tests', 'golden', 'weights_diffuse_pairwise_h256_energy.npz')))
    gd = GaussianDiffusion(den, timesteps=1000, EBM='MALA', samples_per_step=10)
    b = worlds.triangular_batch(int(sys.argv[3]), 12, seed=5).to_torch(dev)
    import time
    x = gd.sample(b, seed=7)
    torch.cuda.synchronize(); t0 = time.time()
    x = gd.sample(b, seed=7)
    torch.cuda.synchronize(); dt = time.time() - t0
    np.savez(sys.argv[2], x=x.cpu().numpy(), acc=gd.last_accept_rates.cpu().numpy(), dt=dt)
else:
    out = []
    g = sys.argv[3] if len(sys.argv) > 3 else '256'
    for k, so in enumerate(sys.argv[1:3]):
        f = '/tmp/cmp_mala_%d.npz' % k
        subprocess.check_call([sys.executable, __file__, '--run', f, g], env=dict(os.environ, CCSP_SO=os.path.abspath(so)))
        out.append(np.load(f))
    a, b = out
    print('poses bitwise equal:', bool(np.array_equal(a['x'], b['x'], equal_nan=True)), ' acceptance rates bitwise equal:', bool(np.array_equal(a['acc'], b['acc'])),
          ' max |dx|', float(np.nanmax(np.abs(a['x'] - b['x']))), ' seconds per chain %.3f / %.3f' % (float(a['dt']), float(b['dt'])),
          ' mean acceptance %.5f' % float(a['acc'].mean()))
