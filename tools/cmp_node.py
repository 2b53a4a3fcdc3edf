import os, sys, subprocess, numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import torch
    from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion, worlds, _lib
    if os.environ.get('CCSP_SO'):
        _lib.SO = os.environ['CCSP_SO']; _lib._stale = lambda *a: False
    from bench import load_weights
    dev = torch.device('cuda:0')
    den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=256, input_mode='qualitative', device=dev, verbose=False)
    den.load_state_dict(load_weights(os.path.join(ROOT, 'tests', 'golden', 'weights_qualitative_h256.npz')))
    gd = GaussianDiffusion(den, timesteps=1000, EBM='ULA', samples_per_step=int(os.environ.get('SPS', '10')))
    b = worlds.qualitative_batch(12, 5, seed=5).to_torch(dev)
    x0 = torch.zeros(b.x.shape[0], 4, device=dev)
    nt = int(sys.argv[2])
    x = gd.p_sample_segment(b, x0, 999, 999 - nt + 1, seed=3)
    np.save(sys.argv[1], x.cpu().numpy())
else:
    for nt in (1, 2):
        for tag in ('generic', 'direct'):
            subprocess.check_call([sys.executable, __file__, '/tmp/x_%s.npy' % tag, str(nt)], env=dict(os.environ, CCSP_NODE=tag, CCSP_LANES='1'))
        a, b = np.load('/tmp/x_generic.npy'), np.load('/tmp/x_direct.npy')
        d = np.abs(a - b)
        print('timesteps', nt, 'max diff', d.max(), 'n differing', int((a != b).sum()), 'of', a.size, 'max |x|', np.abs(a).max())
    import numpy as np
    from diffusion_ccsp_amd import worlds
    bb = worlds.qualitative_batch(12, 5, seed=5)
    a, b = np.load('/tmp/x_generic.npy'), np.load('/tmp/x_direct.npy')
    deg = np.bincount(np.concatenate([bb.edge_index[0], bb.edge_index[1]]), minlength=a.shape[0])
    rows, cols = np.nonzero(a != b)
    print('differing (node, col, degree, mask):', [(int(r), int(c), int(deg[r]), int(bb.mask[r])) for r, c in zip(rows, cols)][:40])
    print('degree histogram of all nodes', np.bincount(deg))
