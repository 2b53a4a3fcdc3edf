"""solved rate of a series of checkpoints (tools/train_gpu.py, CKPT_KSTEPS) the way the reference accounts for it -- Trainer.evaluate
(ddpm.py:558-843, mirrored by diffusion-ccsp_amd/evaluate.py): test sets of 100 graphs per object count, tries=(10, 0), top-1 / top-10 --
plus the share of graphs whose poses end non-finite on a one-try 256-graph batch.
usage (GPU box): python tools/solved_curve.py <out.json> <ckpt.npz> [<ckpt.npz> ...]"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import load_weights
from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion, checker, evaluate, worlds

dev = torch.device('cuda:0')
objects = [int(v) for v in os.environ.get('SWEEP_OBJECTS', '2,3,4,5,6,8').split(',')]
# SPS = ULA steps per timestep (the GaussianDiffusion default is 10; the reference's train_ddpm.py passes 3, train_ddpm.py:35); SAMPLER=none: EBM False
SPS = int(os.environ.get('SPS', '10'))
EBM = False if os.environ.get('SAMPLER', 'ULA') == 'none' else os.environ.get('SAMPLER', 'ULA')
rng = np.random.default_rng(11)
sets = {}
for n_obj in objects:
    gs = []
    for _ in range(100):
        wd = worlds.sample_qualitative_world(rng, n_obj)
        gs.append(worlds.encode_qualitative(wd['nodes'], wd['constraints']))
    sets[n_obj] = gs
out = {}
for path in sys.argv[2:]:
    den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=256, input_mode='qualitative', EBM='ULA', device=dev, verbose=False)
    den.load_state_dict(load_weights(path))
    gd = GaussianDiffusion(den, timesteps=1000, EBM=EBM, samples_per_step=SPS)
    with tempfile.TemporaryDirectory() as td:
        log = evaluate.Evaluator(gd, sets, td).evaluate(0, tries=(10, 0), run_all=True, seed=500)
    rec = {'evaluate': {str(k): {'top1': v['success_rate'], 'top10': v.get('success_rate_top10', v.get('success_rate_top3'))} for k, v in log.items()}}
    one = {}
    for n_obj in objects:
        b = worlds.qualitative_batch(256, n_obj, seed=11 + n_obj)
        x = gd.sample(b.to_torch(dev), seed=3).cpu().numpy()
        ok = checker.solved_mask(x, b)
        bad = len(set(np.asarray(b.batch)[~np.isfinite(x).all(axis=1)].tolist()))
        one[str(n_obj)] = {'solved_of_256': int(ok.sum()), 'nonfinite_graphs_of_256': bad}
    rec['one_try'] = one
    out[os.path.basename(path)] = rec
    print(os.path.basename(path), json.dumps(rec), flush=True)
json.dump(out, open(sys.argv[1], 'w'), indent=1)
