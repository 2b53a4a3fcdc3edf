import os, sys, subprocess, numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import torch
    from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion, worlds
    from bench import load_weights
    dev = torch.device('cuda:0')
    den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=256, input_mode='qualitative', device=dev, verbose=False)
    den.load_state_dict(load_weights(os.path.join(ROOT, 'tests', 'golden', 'weights_qualitative_h256.npz')))
    gd = GaussianDiffusion(den, timesteps=1000, EBM='ULA', samples_per_step=0)
    b = worlds.qualitative_batch(12, 5, seed=5).to_torch(dev)
    N = b.x.shape[0]
    x0 = (torch.arange(N * 4, device=dev, dtype=torch.float32).reshape(N, 4) % 7) * 0.1 - 0.3
    nz = torch.zeros(3, N, 4, device=dev)
    x = gd.p_sample_segment(b, x0, 999, 999, noise=(nz,))
    eps = den(x0, b, torch.tensor([999]), eval=True)
    np.savez(sys.argv[1], x=x.cpu().numpy(), eps=eps.cpu().numpy(), x0=x0.cpu().numpy(),
             a=float(gd.sqrt_recip_alphas_cumprod[999]), b=float(gd.sqrt_recipm1_alphas_cumprod[999]), c1=float(gd.posterior_mean_coef1[999]), c2=float(gd.posterior_mean_coef2[999]))
else:
    for tag in ('generic', 'direct'):
        subprocess.check_call([sys.executable, __file__, '/tmp/x_%s.npz' % tag], env=dict(os.environ, CCSP_NODE=tag, CCSP_LANES='1'))
    A, B = np.load('/tmp/x_generic.npz'), np.load('/tmp/x_direct.npz')
    f = np.float32
    eps, x0 = A['eps'], A['x0']
    assert np.array_equal(A['eps'], B['eps'])
    xh = (f(A['c1']) * (f(A['a']) * x0 - f(A['b']) * eps).astype(f)).astype(f)
    xh = (xh + (f(A['c2']) * x0).astype(f)).astype(f)
    m = np.zeros(x0.shape[0], bool); m[::6] = True
    print('generic == host (free rows):', int((A['x'][~m] != xh[~m]).sum()), ' direct == host:', int((B['x'][~m] != xh[~m]).sum()), ' generic vs direct:', int((A['x'] != B['x']).sum()))
