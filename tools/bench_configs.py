"""Throughput of the BASELINE.json configurations other than the headline one, per GPU, synthetic graphs and
plumbing weights (reset_parameters) -- a timing of the kernels, not a quality statement.  Numbers go to DESIGN.md.
usage: python tools/bench_configs.py [c4|c5|sd|c1 ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diffusion_ccsp_amd import ComposedEBMDenoiseFn, ConstraintDiffuser, GaussianDiffusion, worlds

dev = torch.device('cuda:0')


def run(tag, mode, batch, EBM, H=256, T=1000, S=10, energy=False, model='Diffusion-CCSP', reps=2):
    den = ConstraintDiffuser(dims=worlds.MODE_DIMS[mode], hidden_dim=H, input_mode=mode, EBM=EBM, energy_wrapper=energy,
                             device=dev, verbose=False, model=model)
    den.reset_parameters(0)
    fn = ComposedEBMDenoiseFn(den) if energy else den
    gd = GaussianDiffusion(fn, timesteps=T, EBM=EBM, samples_per_step=S)
    b = batch.to_torch(dev)
    n_graphs = int(batch.batch.max()) + 1
    gd.sample(b, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        x = gd.sample(b, seed=2 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('%-4s %-52s graphs %4d nodes %5d edges %6d  %8.1f ms/chain  %8.1f samples/s  evals %d' %
          (tag, '%s %s T=%d S=%s H=%d %s' % (mode, EBM, T, S, H, model), n_graphs, b.x.shape[0], b.edge_index.shape[1],
           1e3 * dt, n_graphs / dt, gd.chain_stats()['evals']), flush=True)


which = set(sys.argv[1:]) or {'c1', 'c4', 'c5', 'sd'}
if 'c1' in which:      # configs[0] shape: 3 objects, T=100, batch 1 (latency of one small chain)
    run('C1', 'qualitative', worlds.qualitative_batch(1, 3, seed=1), 'ULA', T=100)
if 'c4' in which:      # configs[3]: triangular 12 objects, MALA, 1024 graphs over 4 GPUs -> 256 per GPU
    run('C4', 'diffuse_pairwise', worlds.triangular_batch(256, 12, seed=2), 'MALA', energy=True, reps=1)
if 'c5' in which:      # configs[4]: panda-box 10 objects, 512 graphs over 8 GPUs -> 64 per GPU
    run('C5', 'robot_box', worlds.robot_box_batch(64, 10, seed=3), 'ULA')
if 'sd' in which:      # StructDiffusion baseline: 7 objects (8 tokens), 256 graphs
    run('SD', 'qualitative', worlds.qualitative_batch(256, 7, seed=4), 'ULA', model='StructDiffusion', reps=1)
