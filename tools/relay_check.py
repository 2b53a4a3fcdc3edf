"""Relay mode (CCSP_RELAY, csrc Gate) against the stream-ordered launches: bitwise equality and chain time on small batches.  tools only.
usage (GPU box): python tools/relay_check.py [T]"""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import diffusion_ccsp_amd
from diffusion_ccsp_amd import _lib
if os.environ.get('CCSP_SO'):
    _lib.SO = os.environ['CCSP_SO']; _lib._stale = lambda *a: False
from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion, worlds
from bench import load_weights

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device('cuda:0')


def model(mode, wfile, relay):
    os.environ['CCSP_RELAY'] = str(relay)
    den = ConstraintDiffuser(dims=worlds.MODE_DIMS[mode], hidden_dim=256, input_mode=mode, device=dev, verbose=False)
    den.load_state_dict(load_weights(os.path.join(ROOT, 'tests', 'golden', wfile)))
    return GaussianDiffusion(den, timesteps=T, EBM='ULA', samples_per_step=10)


def chain_time(gd, b, reps=5):
    for _ in range(2):
        gd.sample(b, seed=1)
    torch.cuda.synchronize()
    ts = []
    for i in range(reps):
        t0 = time.perf_counter()
        x = gd.sample(b, seed=7)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts), x, gd.chain_stats()['evals']


cases = [('qualitative', 'weights_qualitative_h256.npz', lambda: worlds.qualitative_batch(1, 3, seed=5), 'C1: 1 graph x 3 objects'),
         ('qualitative', 'weights_qualitative_h256.npz', lambda: worlds.qualitative_batch(8, 8, seed=5), '8 graphs x 8 objects'),
         ('qualitative', 'weights_qualitative_h256.npz', lambda: worlds.qualitative_batch(32, 8, seed=5), '32 graphs x 8 objects'),
         ('robot_box', 'weights_robot_box_h256.npz', lambda: worlds.robot_box_batch(64, 10, seed=5), 'C5: 64 graphs x 10 objects')]
for mode, wfile, mk, label in cases:
    b = mk().to_torch(dev)
    res = {}
    for relay in (0, 1):
        gd = model(mode, wfile, relay)
        res[relay] = chain_time(gd, b)
    (t0, x0, ev), (t1, x1, _) = res[0], res[1]
    same = torch.equal(x0, x1) or bool(((x0 == x1) | (x0.isnan() & x1.isnan())).all())
    print('%-28s evaluations %5d  stream-ordered %7.2f us/evaluation  relay %7.2f  (%+.1f %%)  bitwise equal: %s  finite: %s' %
          (label, ev, 1e6 * t0 / ev, 1e6 * t1 / ev, 100.0 * (t0 / t1 - 1.0), same, bool(torch.isfinite(x1).all())), flush=True)
