"""what the checker reports for unsolved samples (collisions vs missing qualitative constraints)"""
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import load_weights
from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion, checker, worlds

dev = torch.device('cuda:0')
den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=256, input_mode='qualitative', EBM='ULA', device=dev, verbose=False)
den.load_state_dict(load_weights(os.path.join(ROOT, 'weights', 'qualitative_h256_ref30k.npz')))
gd = GaussianDiffusion(den, timesteps=1000, EBM='ULA', samples_per_step=10)
for n in (3, 8):
    b = worlds.qualitative_batch(128, n, seed=11 + n)
    x = gd.sample(b.to_torch(dev), seed=3).cpu().numpy().clip(-1, 1)
    gid = np.asarray(b.batch); ei = np.asarray(b.edge_index); ea = np.asarray(b.edge_attr)
    kinds, ncol, nmiss, gt_ok = Counter(), [], [], 0
    for j in range(128):
        nodes = np.nonzero(gid == j)[0]; n0 = int(nodes[0])
        sel = np.nonzero(gid[ei[0]] == j)[0]
        given = [(worlds.QUALITATIVE_CONSTRAINTS[int(ea[e])], int(ei[0, e]) - n0, int(ei[1, e]) - n0) for e in sel]
        feats = np.concatenate([np.asarray(b.x)[nodes, :2], x[nodes]], axis=1)
        ev = checker.evaluate_graph(feats, b.world_dims[j], given)
        gt = checker.evaluate_graph(np.asarray(b.x)[nodes, :6], b.world_dims[j], given)
        gt_ok += len(gt) == 0
        if not ev:
            kinds['solved'] += 1
        elif ev[0][0] in ('north', 'south', 'east', 'west') or str(ev[0][0]).startswith('tile_'):
            kinds['collision'] += 1; ncol.append(len(ev))
        else:
            kinds['missing'] += 1; nmiss.append(len(ev))
            for c in ev: kinds['miss:' + c[0]] += 1
    print(n, 'objects: ground truth passes %d/128;' % gt_ok, dict(kinds), 'mean #collisions %.1f' % (np.mean(ncol) if ncol else 0),
          'mean #missing %.1f' % (np.mean(nmiss) if nmiss else 0), flush=True)
    # pose error against ground truth for context
    gtp = np.asarray(b.x)[:, 2:6]
    print('   mean |x - gt| per column', np.abs(x - gtp).mean(axis=0).round(3))
