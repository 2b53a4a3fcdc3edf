"""TEST INFRASTRUCTURE ONLY (build container; needs /root/reference).

Cross-check of tools/train_gpu.py -- the plain-torch restatement of the training forward that makes bench.py's C2 weights on
the GPU box -- against the reference's own GaussianDiffusion.p_losses (networks/ddpm.py:363-389), which oracle/ref_train.py
trains the parity fixtures with.  Same initial weights (the reference module's state_dict copied into the restatement), same
batches, same (t, noise) draws, both on PyTorch-CPU:
  1. loss and the gradient of EVERY parameter on single batches (relative difference);
  2. 40 Adam steps (lr 5e-4, the reference recipe) side by side: loss curves and final weights;
  3. the two trained networks evaluated on a fresh batch.
usage: python oracle/check_train_gpu.py [--hidden 64]            -> profiles/r02_check_train_gpu.txt
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--hidden', type=int, default=64)
    ap.add_argument('--steps', type=int, default=40)
    args = ap.parse_args()
    os.environ['TRAIN_DEVICE'] = 'cpu'
    os.environ['TRAIN_HIDDEN'] = str(args.hidden)
    import ref_import
    ddpm, dfn = ref_import.load()
    import diffusion_ccsp_amd  # noqa: F401
    from diffusion_ccsp_amd import worlds
    import train_gpu as tg

    torch.manual_seed(3)
    ref_den = dfn.ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=args.hidden, EBM='ULA', input_mode='qualitative',
                                     device='cpu', verbose=False)
    ref_gd = ddpm.GaussianDiffusion(ref_den, timesteps=1000, EBM='ULA', samples_per_step=10, step_sizes='2*self.betas')
    ref_gd.train()
    net = tg.Denoiser()
    missing = net.load_state_dict({k: v.detach().clone() for k, v in ref_den.state_dict().items()}, strict=True)
    sa, sb = tg.schedule_tables()
    rng = np.random.default_rng(0)
    batches = []
    for _ in range(6):
        gs = []
        for _ in range(32):
            wd = worlds.sample_qualitative_world(rng, int(rng.integers(2, 9)))
            gs.append(worlds.encode_qualitative(wd['nodes'], wd['constraints']))
        batches.append(worlds.collate(gs))
    lines = ['check_train_gpu: hidden_dim %d, torch %s' % (args.hidden, torch.__version__)]
    gen = torch.Generator().manual_seed(11)

    def draws(b):
        t = torch.randint(0, 1000, (1,), generator=gen)
        noise = torch.randn((b.x.shape[0], 4), generator=gen)
        return t, noise

    # 1. loss + gradients on single batches
    worst = 0.0
    for b in batches[:3]:
        t, noise = draws(b)
        bt = b.to_torch()
        m = bt.mask.bool()
        nz = noise.clone()
        nz[m] = 0                                              # conditional_noise (ddpm.py:114-117)
        ref_gd.zero_grad()
        l_ref = ref_gd.p_losses(bt, t, noise=nz, debug=False)
        l_ref.backward()
        net.zero_grad()
        l_new = tg.loss_on(net, tg.to_dev(b), t, noise, sa, sb)
        l_new.backward()
        gmax = 0.0
        for (k, p), (k2, q) in zip(ref_den.named_parameters(), net.named_parameters()):
            assert k == k2, (k, k2)
            d = float((p.grad - q.grad).abs().max() / (1e-12 + p.grad.abs().max()))
            gmax = max(gmax, d)
        worst = max(worst, gmax, abs(float(l_ref) - float(l_new)) / abs(float(l_ref)))
        lines.append('  batch: t=%4d  loss reference %.7f  restatement %.7f   max relative gradient difference over %d parameters %.2e'
                     % (int(t), float(l_ref), float(l_new), len(list(net.parameters())), gmax))
    # 2. Adam side by side
    o_ref = torch.optim.Adam(ref_gd.parameters(), lr=5e-4)
    o_new = torch.optim.Adam(net.parameters(), lr=5e-4)
    curve = []
    for step in range(args.steps):
        b = batches[step % len(batches)]
        t, noise = draws(b)
        bt = b.to_torch()
        nz = noise.clone()
        nz[bt.mask.bool()] = 0
        l_ref = ref_gd.p_losses(bt, t, noise=nz, debug=False)
        o_ref.zero_grad(); l_ref.backward(); o_ref.step()
        l_new = tg.loss_on(net, tg.to_dev(b), t, noise, sa, sb)
        o_new.zero_grad(); l_new.backward(); o_new.step()
        curve.append((float(l_ref), float(l_new)))
    dl = max(abs(a - c) / abs(a) for a, c in curve)
    dw = max(float((p - q).abs().max() / (1e-12 + p.abs().max())) for p, q in zip(ref_den.parameters(), net.parameters()))
    lines.append('  %d Adam steps side by side: loss %.5f -> %.5f (reference), %.5f -> %.5f (restatement); max relative loss difference %.2e, '
                 'max relative weight difference afterwards %.2e' % (args.steps, curve[0][0], curve[-1][0], curve[0][1], curve[-1][1], dl, dw))
    # 3. the two trained networks on a fresh batch
    wd_b = worlds.qualitative_batch(4, 8, seed=77)
    bt = wd_b.to_torch()
    poses = torch.randn((bt.x.shape[0], 4), generator=gen) * 0.6
    ref_den.eval()
    with torch.no_grad():
        o1 = ref_den(poses.clone(), bt, torch.tensor([500]), eval=True)
        o2 = net(poses.clone(), tg.to_dev(wd_b), torch.tensor([500]))
    do = float((o1 - o2).abs().max() / (1 + o1.abs().max()))
    lines.append('  evaluation of both trained networks on a fresh 4 x 8-object batch: max relative output difference %.2e' % do)
    ok = worst < 1e-4 and dl < 1e-3 and do < 1e-4
    lines.append('  %s' % ('OK' if ok else 'FAIL'))
    out = '\n'.join(lines)
    print(out)
    with open(os.path.join(ROOT, 'profiles', 'r02_check_train_gpu.txt'), 'w') as f:
        f.write(out + '\n')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
