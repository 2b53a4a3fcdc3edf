"""TEST / BASELINE INFRASTRUCTURE ONLY (build container; needs /root/reference).

Certifies oracle/torch_proxy.py as a stand-in for "the reference's PyTorch-CPU path" (SURVEY.md 8c-2): on the same batch,
with the same weights, the same thread count and interleaved repetitions, the imported reference and the proxy must
  * give the same outputs (<= 1e-6 relative; observed: bit-identical), and
  * cost the same per network evaluation within +-5 %.
The direct mode (C2-shaped batch: 256 x 8-object qualitative graphs), the energy mode (C4-shaped: 12-triangle graphs,
forward + autograd backward) and, since round 6, the StructDiffusion transformer baseline (256 x 7-object graphs, width 512) are checked.  The GPU box cannot run the reference (only /root/repo travels), so bench.py's
cpu_baseline times the proxy there; this script is what ties that number to the reference.

usage: python oracle/certify_proxy.py [--graphs 256] [--threads 8] [--reps 5] [--out profiles/r06_certify_proxy.txt]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402
import oracle as oracle_mod  # noqa: E402
import torch_proxy  # noqa: E402
import diffusion_ccsp_amd  # noqa: E402,F401
from diffusion_ccsp_amd import worlds  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def reference_model(dfn, mode, H, W, energy, model='Diffusion-CCSP'):
    m = dfn.ConstraintDiffuser(dims=worlds.MODE_DIMS[mode], hidden_dim=H, EBM='MALA' if energy else 'ULA', input_mode=mode,
                               energy_wrapper=energy, device='cpu', verbose=False, model=model)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    return m.eval()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--graphs', type=int, default=256)
    ap.add_argument('--threads', type=int, default=8)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r06_certify_proxy.txt'))
    args = ap.parse_args()
    ddpm, dfn = ref_import.load()
    torch.set_num_threads(args.threads)
    lines = ['certify_proxy: torch %s, %d threads, %d interleaved repetitions per arm' % (torch.__version__, args.threads, args.reps)]
    ok = True
    cases = [('direct  C2-shaped', 'qualitative', 13, 'weights_qualitative_h256.npz', False, worlds.qualitative_batch(args.graphs, 8, seed=5), 'Diffusion-CCSP'),
             ('energy  C4-shaped', 'diffuse_pairwise', 2, 'weights_diffuse_pairwise_h256_energy.npz', True,
              worlds.triangular_batch(max(1, args.graphs // 4), 12, seed=5), 'Diffusion-CCSP'),
             # round 6: the transformer baseline (bench.py --config sd: 256 graphs x 7 objects = 8-token sequences, width 512)
             ('transformer baseline', 'qualitative', 13, 'weights_qualitative_h256_sd.npz', False, worlds.qualitative_batch(args.graphs, 7, seed=5), 'StructDiffusion')]
    for tag, mode, C, wfile, energy, batch, kind in cases:
        W = oracle_mod.load_weights(os.path.join(GOLD, wfile))
        ref = reference_model(dfn, mode, 256, W, energy, kind)
        prx = (torch_proxy.ProxyStructDiffuser if kind == 'StructDiffusion' else torch_proxy.ProxyDiffuser)(W, worlds.MODE_DIMS[mode], 256, C)
        b = batch.to_torch()
        N, P = b.x.shape[0], worlds.MODE_DIMS[mode][-1][0]
        poses = (torch.randn(N, P, generator=torch.Generator().manual_seed(1)) * 0.6)
        t = torch.tensor([417])

        def run_ref():
            if energy:
                g, e = ref(poses.clone().requires_grad_(True), b, t, eval=True, tag='EBM')
                return g.detach()
            with torch.no_grad():
                return ref(poses.clone(), b, t, eval=True)

        def run_prx():
            if energy:
                return prx.energy_grad(poses, b, t)[0]
            with torch.no_grad():
                return prx(poses.clone(), b, t)
        a, c = run_ref(), run_prx()                         # warm-up + outputs
        err = float((a - c).abs().max() / (1 + a.abs().max()))
        tr, tp = [], []
        for k in range(args.reps):                          # alternate which arm goes first (cache / allocator state)
            for arm in ((run_ref, tr), (run_prx, tp)) if k % 2 == 0 else ((run_prx, tp), (run_ref, tr)):
                t0 = time.perf_counter()
                arm[0]()
                arm[1].append(time.perf_counter() - t0)
        mr, mp = float(np.min(tr)), float(np.min(tp))              # (minimum: the container shares its cores)
        good = err <= 1e-6 and abs(mp / mr - 1) <= 0.05
        ok &= good
        lines.append('%s  %4d graphs  %6d edges   reference %8.2f ms/eval   proxy %8.2f ms/eval   ratio %.3f   |out diff| %.1e   %s'
                     % (tag, int(np.asarray(batch.batch).max()) + 1, b.edge_index.shape[1], 1e3 * mr, 1e3 * mp, mp / mr, err, 'OK' if good else 'FAIL'))
    out = '\n'.join(lines)
    print(out)
    os.makedirs(os.path.join(ROOT, 'profiles'), exist_ok=True)
    with open(args.out, 'w') as f:
        f.write(out + '\n')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
