"""TEST INFRASTRUCTURE ONLY (build container; needs /root/reference).

Generates the golden vectors under tests/golden/ by running the *reference itself*
(zt-yang/diffusion-ccsp, imported unmodified through oracle/ref_import.py) on PyTorch-CPU.  The
reference holds no tests, fixtures or golden vectors for the sampling path, so these
files are what pins the oracle (oracle/ccsp_oracle.c) and, through it and directly, the HIP path.

Only arrays are written: inputs (graphs, poses, noise seeds), expected outputs, and the weights
trained by oracle/ref_train.py.  No reference source or bytecode is copied.

Noise: ``torch.randn`` / ``torch.rand`` are temporarily replaced by the build-owned counter-based
stream (diffusion-ccsp_amd/noise.py), because networks/ddpm.py calls them as module attributes
(ddpm.py:121-122,255,273,292,1037).

usage: python oracle/gen_golden.py [names...]      (no names = everything)
"""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402
import oracle as oracle_mod  # noqa: E402  (only for load_weights)
import diffusion_ccsp_amd  # noqa: E402,F401
from diffusion_ccsp_amd import noise, worlds  # noqa: E402

ddpm, dfn = ref_import.load()

# AnnealedULASampler.sample_step runs under @torch.enable_grad() (ddpm.py:955), so the state it
# returns carries the autograd graph of its S network evaluations; with return_history=True the
# reference keeps all T of them alive (tens of GB at hidden_dim 256).  Detaching the returned state
# changes no value -- p_sample runs under no_grad anyway -- and lets the capture fit in memory.
_ula_step = ddpm.AnnealedULASampler.sample_step
ddpm.AnnealedULASampler.sample_step = lambda self, x, batch, t: _ula_step(self, x, batch, t).detach()


class PatchedNoise(object):
    def __init__(self, seed, dtype=torch.float32):
        self.seed, self.c, self.uc, self.dtype = seed, 0, 0, dtype

    @staticmethod
    def _shape(shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            return tuple(shape[0])
        return tuple(shape)

    def __enter__(self):
        self.orig = (torch.randn, torch.rand, torch.randn_like)

        def randn(*shape, **kw):
            shape = self._shape(shape)
            z = noise.normal(self.seed, self.c, shape[0], shape[1])
            self.c += 1
            return torch.from_numpy(z).to(self.dtype)

        def rand(*shape, **kw):
            shape = self._shape(shape)
            u = noise.uniform(self.seed, self.uc, shape[0])
            self.uc += 1
            return torch.from_numpy(u).to(self.dtype)

        # AnnealedMUHASampler draws with torch.randn_like (ddpm.py:1090,1096): same stream, same call counter
        torch.randn, torch.rand, torch.randn_like = randn, rand, (lambda t, **kw: randn(*t.shape))
        return self

    def __exit__(self, *a):
        torch.randn, torch.rand, torch.randn_like = self.orig


def build_reference(mode, H, weights, energy=False, EBM='ULA', T=1000, S=10, dtype=torch.float32,
                    model_name='Diffusion-CCSP', ebm_per_steps=1):
    dims = worlds.MODE_DIMS[mode]
    if dtype == torch.float64:
        torch.set_default_dtype(torch.float64)
    try:
        model = dfn.ConstraintDiffuser(dims=dims, hidden_dim=H, EBM=EBM, input_mode=worlds.ref_mode(mode), energy_wrapper=energy,
                                       device='cpu', verbose=False, model=model_name)
        model.load_state_dict({k: torch.from_numpy(v).to(dtype) for k, v in weights.items()})
        den = dfn.ComposedEBMDenoiseFn(model, ebm_per_steps) if energy else model
        gd = ddpm.GaussianDiffusion(den, timesteps=T, EBM=EBM, samples_per_step=S, step_sizes='2*self.betas')
        if dtype == torch.float64:
            gd = gd.double()
            gd._sqrt_recipm1_alphas_cumprod_custom = gd._sqrt_recipm1_alphas_cumprod_custom.double()
            gd.step_sizes = gd.step_sizes.double() if torch.is_tensor(gd.step_sizes) else gd.step_sizes
    finally:
        torch.set_default_dtype(torch.float32)
    return model, gd.eval()


def batch_arrays(b):
    return dict(x=b.x.numpy().astype(np.float32), edge_index=b.edge_index.numpy().astype(np.int64),
                edge_attr=b.edge_attr.numpy().astype(np.float32), mask=b.mask.numpy().astype(np.int8))


GRASP_KEYS = ['x+', 'x-', 'y+', 'y-', 'z+']
HIST_IDX = [0, 1, 2, 3, 4, 5, 10, 50, 100, 200, 300, 400, 500, 600, 700, 800, 900, 950, 990, 998, 999, 1000]


def run_chain(name, mode, H, wfile, batch, EBM, T=1000, S=10, seed=7, energy=False, dtype=torch.float32,
              model_name='Diffusion-CCSP', ebm_per_steps=1, full_hist=False):
    W = oracle_mod.load_weights(os.path.join(ROOT, wfile) if os.sep in wfile else os.path.join(GOLD, wfile))
    model, gd = build_reference(mode, H, W, energy=energy, EBM=EBM, T=T, S=S, dtype=dtype, model_name=model_name,
                                ebm_per_steps=ebm_per_steps)
    b = batch.clone()
    if dtype == torch.float64:
        b.x = b.x.double()
    t0 = time.time()
    torch.set_default_dtype(dtype)          # SinusoidalPosEmb builds its table in the default dtype
    rates = {}
    orig_upd = ddpm.MetropolisSampler._update_acceptance_rate

    def record(self, accept_rate, t, debug=False):          # mean acceptance of the timestep, as the reference logs it
        rates[int(t)] = float(accept_rate)
        return orig_upd(self, accept_rate, t, debug)
    ddpm.MetropolisSampler._update_acceptance_rate = record
    try:
        with PatchedNoise(seed, dtype) as pn, contextlib.redirect_stdout(io.StringIO()):
            out, hist = gd.sample(b, return_history=True)
    finally:
        torch.set_default_dtype(torch.float32)
        ddpm.MetropolisSampler._update_acceptance_rate = orig_upd
    dt = time.time() - t0
    out = out.detach().numpy()
    hist = np.stack([h.detach().numpy() for h in hist])
    idx = list(range(T + 1)) if (T <= 20 or full_hist) else sorted(set(i for i in HIST_IDX if i <= T) | {T})
    rec = dict(batch_arrays(batch))
    if model_name == 'StructDiffusion':
        rec['batch'] = batch.batch.numpy().astype(np.int64)
        if hasattr(batch, 'shuffled'):
            rec['shuffled'] = batch.shuffled.numpy().astype(np.int64)
    rec.update(final=out.astype(np.float64 if dtype == torch.float64 else np.float32),
               hist_idx=np.asarray(idx, dtype=np.int32), hist=hist[idx].astype(out.dtype),
               seed=np.int64(seed), T=np.int32(T), S=np.int32(S), H=np.int32(H), n_randn=np.int64(pn.c),
               n_rand=np.int64(pn.uc), ref_seconds=np.float64(dt), threads=np.int32(torch.get_num_threads()))
    if rates:
        rec['accept'] = np.asarray([rates.get(t, 0.0) for t in range(T)], dtype=np.float32)
    meta = dict(mode=mode, EBM=str(EBM), weights=wfile, energy=bool(energy), dtype=str(dtype), model=model_name,
                ebm_per_steps=int(ebm_per_steps))
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), meta=np.asarray(repr(meta)), **rec)
    print('%-34s %6.1fs  randn calls %d  |final|max %.3f  |hist|max %.3g' %
          (name, dt, pn.c, np.abs(out).max(), np.abs(hist).max()), flush=True)


def gen_schedule():
    rec = {}
    for T in (100, 1000):
        W = oracle_mod.load_weights(os.path.join(GOLD, 'weights_qualitative_h64.npz'))
        _, gd = build_reference('qualitative', 64, W, T=T)
        for k in ['betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_recip_alphas_cumprod',
                  'sqrt_recipm1_alphas_cumprod', 'posterior_log_variance_clipped', 'posterior_mean_coef1',
                  'posterior_mean_coef2', 'posterior_variance', 'sqrt_alphas_cumprod', 'sqrt_one_minus_alphas_cumprod',
                  'log_one_minus_alphas_cumprod']:
            rec['T%d/%s' % (T, k)] = getattr(gd, k).numpy()
        # key names of the reference module's state_dict (what Trainer.save writes, ddpm.py:496-501), in order
        rec['T%d/state_dict_keys' % T] = np.asarray(list(gd.state_dict().keys()))
        rec['T%d/kappa' % T] = gd._sqrt_recipm1_alphas_cumprod_custom.numpy()
        rec['T%d/step_sizes' % T] = gd.step_sizes.numpy()
    np.savez_compressed(os.path.join(GOLD, 'schedule.npz'), **rec)
    print('schedule.npz')


def gen_operators():
    """the denoiser's sub-modules called the way visualize_energy.py:402-450 calls them: geom_encoder, pose_encoder,
    time_mlp on a FLOAT timestep, _process_constraint(i, input_dict) on a grid of pose pairs (+ the grasp branch)"""
    rec = {}
    rng = np.random.default_rng(55)
    for tag, mode, wfile in (('q64', 'qualitative', 'weights_qualitative_h64.npz'), ('r64', 'robot_box', 'weights_robot_box_h64.npz'),
                             ('q256', 'qualitative', 'weights_qualitative_h256.npz')):
        W = oracle_mod.load_weights(os.path.join(GOLD, wfile))
        H = int(wfile.split('_h')[1].split('.')[0])
        model, _ = build_reference(mode, H, W)
        dims = worlds.MODE_DIMS[mode]
        n = 12
        geoms_in = rng.uniform(0.05, 0.9, (2, dims[0][0])).astype(np.float32)
        poses_in = rng.uniform(-1, 1, (n, 2, dims[-1][0])).astype(np.float32)
        tval = np.asarray([417.25], dtype=np.float32)
        with torch.no_grad():
            ge = model.geom_encoder(torch.from_numpy(geoms_in))
            te = model.time_mlp(torch.from_numpy(tval))
            pe = model.pose_encoder(torch.from_numpy(poses_in))
            d = {'args': None, 'geoms_emb': ge[None].repeat(n, 1, 1), 'poses_emb': pe, 'time_embedding': te.repeat(n, 1)}
            if 'robot' in mode:
                grasp_in = np.eye(5, dtype=np.float32)[rng.integers(0, 5, n)]
                d['grasp_emb'] = model.grasp_encoder(torch.from_numpy(grasp_in))
                rec[tag + '/grasp_in'] = grasp_in
                rec[tag + '/grasp_emb'] = d['grasp_emb'].numpy()
            outs = np.stack([model._process_constraint(i, d).numpy() for i in range(len(model.mlps))])
        rec.update({tag + '/geoms_in': geoms_in, tag + '/poses_in': poses_in, tag + '/t': tval, tag + '/geoms_emb': ge.numpy(),
                    tag + '/poses_emb': pe.numpy(), tag + '/time_emb': te.numpy(), tag + '/outputs': outs})
    np.savez_compressed(os.path.join(GOLD, 'operators.npz'), **rec)
    print('operators.npz')


def gen_evaluate_summary():
    """Trainer.summarize_success_rate (ddpm.py:823-843) -- the success accounting and the log record of Trainer.evaluate --
    called on scripted bookkeeping states through the reference's own method (wandb stubbed, use_wandb False)"""
    import json
    import types
    sys.modules.setdefault('wandb', types.ModuleType('wandb'))
    rng = np.random.default_rng(17)
    cases = []
    for c in range(12):
        count = int(rng.integers(3, 40))
        tries = int(rng.integers(1, 6))
        success_list, success_rounds, succeeded = [], {}, []
        for j in range(count):
            for k in range(tries):
                if j in success_rounds and rng.random() < 0.5:
                    continue                                    # (first-loader graphs are not re-checked once solved; later loaders may)
                if rng.random() < 0.25:
                    success_list.append((j, k))
                    if j not in success_rounds:
                        succeeded.append(j)
                        success_rounds[j] = k
        times = [float(v) for v in rng.uniform(0.2, 3.0, int(rng.integers(1, 11)))]
        fake = types.SimpleNamespace(model=types.SimpleNamespace(sample_loop_time=list(times)), use_wandb=False)
        log = {}
        final = bool(c % 2)
        with contextlib.redirect_stdout(io.StringIO()):
            ddpm.Trainer.summarize_success_rate(fake, count, list(success_list), count, list(succeeded), dict(success_rounds), log,
                                                tries=tries, send_wandb=final)
        cases.append(dict(i=count, count=count, tries=tries, success_list=success_list, succeeded=succeeded,
                          success_rounds={str(k): v for k, v in success_rounds.items()}, sample_loop_time=times, final=final,
                          expected=json.loads(json.dumps(log[count])), sample_loop_time_after=list(fake.model.sample_loop_time)))
    with open(os.path.join(GOLD, 'evaluate_summary.json'), 'w') as f:
        json.dump(cases, f)
    print('evaluate_summary.json')


def gen_labeller():
    """objects -> constraints of the reference's qualitative labeller (envs/data_utils.py:427-621)"""
    _, du = ref_import.load_envs()
    rng = np.random.default_rng(123)
    rec = {}
    for i in range(40):
        n = int(rng.integers(2, 9))
        wd = worlds.sample_qualitative_world(rng, n)
        objs = {k: dict(label=k, center=v['center'], extents=v['extents']) for k, v in wd['objects'].items()}
        ref = du.compute_qualitative_constraints(objs, rotations={}, scale=1)
        boxes = np.asarray([list(v['center'][:2]) + list(v['extents'][:2]) for k, v in wd['objects'].items() if k.startswith('tile_')])
        rec['case%d/boxes' % i] = boxes
        rec['case%d/cons' % i] = np.asarray([[worlds.QUALITATIVE_CONSTRAINTS.index(c[0]), c[1], c[2]] for c in ref], dtype=np.int32).reshape(-1, 3)
    np.savez_compressed(os.path.join(GOLD, 'labeller.npz'), **rec)
    print('labeller.npz')


def gen_pre_transform():
    """raw graphs -> normalised Data rows by the reference's data_transform_cn_diffuse_batch
    (networks/data_transforms.py:26-200) and stability_data_json_to_pt (:272-303)"""
    import data_transforms as dt
    from torch_geometric.data import Data
    rng = np.random.default_rng(7)
    rec = {}
    names = {'qualitative': worlds.QUALITATIVE_CONSTRAINTS, 'diffuse_pairwise': worlds.PUZZLE_CONSTRAINTS,
             'stability_flat': worlds.STABILITY_CONSTRAINTS, 'robot_box': worlds.ROBOT_CONSTRAINTS}

    def run(tag, raw_x, raw_edges, mode):
        d = Data(x=torch.tensor(raw_x, dtype=torch.float), edge_index=[tuple(e) for e in raw_edges], y=None)
        out = dt.data_transform_cn_diffuse_batch(d, 0, mode)[0]
        rec[tag + '/mode'] = np.asarray(mode)
        rec[tag + '/raw_x'] = np.asarray(raw_x, dtype=np.float64)
        rec[tag + '/raw_edges'] = np.asarray([[names[mode].index(e[0]), e[1], e[2]] for e in raw_edges], dtype=np.int64)
        rec[tag + '/x'] = out.x.numpy()
        rec[tag + '/edge_index'] = out.edge_index.numpy()
        rec[tag + '/edge_attr'] = out.edge_attr.numpy()
        rec[tag + '/mask'] = out.mask.numpy()
        rec[tag + '/world_dims'] = np.asarray(out.world_dims, dtype=np.float64)

    for i in range(3):                                       # qualitative worlds from the generator
        wd = worlds.sample_qualitative_world(rng, int(rng.integers(2, 7)))
        run('q%d' % i, wd['nodes'], wd['constraints'], 'qualitative')
    # triangle P1 with sin/cos (8 columns) and with theta (7 columns), plain boxes (5 columns)
    n = 5
    tri8 = [[0, 3, 3, 0, 0, 0, 0, 0]] + [[1] + rng.uniform(0.2, 1.0, 3).tolist() + rng.uniform(-1, 1, 2).tolist()
                                           + [np.cos(a), np.sin(a)] for a in rng.uniform(-3, 3, n)]
    edges = [('in', i, 0) for i in range(1, n + 1)] + [('cfree', i, j) for i in range(1, n) for j in range(i + 1, n + 1)]
    run('tri8', tri8, edges, 'diffuse_pairwise')
    tri7 = [[0, 3, 3, 0, 0, 0, 0]] + [[1] + rng.uniform(0.2, 1.0, 3).tolist() + rng.uniform(-1, 1, 2).tolist() + [a]
                                       for a in rng.uniform(-3, 3, n)]
    run('tri7', tri7, edges, 'diffuse_pairwise')
    box5 = [[0, 3, 2, 0, 0]] + [[1] + rng.uniform(0.2, 1.0, 2).tolist() + rng.uniform(-1, 1, 2).tolist() for _ in range(n)]
    run('box5', box5, edges, 'diffuse_pairwise')
    # stability: json -> raw graph -> Data
    container = dict(shelf_extent=[1.2, 0.8, 0.02], shelf_pose=[0.3, -0.1, 0.5])
    placements = [dict(extents=rng.uniform(0.1, 0.4, 3).tolist(), centroid=(rng.uniform(-0.4, 0.4, 3) + [0.3, -0.1, 0.6]).tolist(),
                       theta=float(rng.uniform(-1.5, 1.5))) for _ in range(4)]
    supports = [(2, 1), (4, 3)]
    sd = dt.stability_data_json_to_pt(dict(container=container, placements=placements, supports=supports), 'x', 'stability_flat')
    rec['stab/container_extent'] = np.asarray(container['shelf_extent'])
    rec['stab/container_pose'] = np.asarray(container['shelf_pose'])
    rec['stab/extents'] = np.asarray([p['extents'] for p in placements])
    rec['stab/centroids'] = np.asarray([p['centroid'] for p in placements])
    rec['stab/thetas'] = np.asarray([p['theta'] for p in placements])
    rec['stab/supports'] = np.asarray(supports, dtype=np.int64)
    rec['stab/ref_raw_x'] = sd.x.numpy()
    rec['stab/ref_raw_edges'] = np.asarray([[worlds.STABILITY_CONSTRAINTS.index(e[0]), e[1], e[2]] for e in sd.edge_index], dtype=np.int64)
    run('stab', sd.x.numpy().astype(np.float64), sd.edge_index, 'stability_flat')
    # robot json -> raw graph by robot_data_json_to_pt (:203-269).  pybullet_planning is absent: the stub supplies only
    # euler_from_quat (pybullet's standard x, y, z, w -> roll, pitch, yaw), and the fixture's place poses are pure yaw
    # rotations, for which every convention-consistent extraction returns the same yaw
    import types

    def euler_from_quat(q):
        x, y, z, w = q
        return (float(np.arctan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y))), float(np.arcsin(np.clip(2 * (w * y - z * x), -1, 1))),
                float(np.arctan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))))
    sys.modules['pybullet_planning'] = types.SimpleNamespace(euler_from_quat=euler_from_quat)
    rcont = dict(tray_dim=[0.42, 0.55, 0.1], tray_pose=[0.5, -0.2, 0.3])
    rplace = []
    for k in range(5):
        yaw = float(rng.uniform(-3, 3))
        rplace.append(dict(name='Bottle_%d' % (3000 + k), extent=rng.uniform(0.04, 0.2, 3).tolist(), scale=float(rng.uniform(0.5, 1.5)),
                           grasp_id=int(rng.integers(0, 12)), grasp_side=[[GRASP_KEYS[int(rng.integers(0, 5))], int(rng.choice([-1, 1]))]],
                           place_pose=[(rng.uniform(-0.15, 0.15, 3) + [0.5, -0.2, 0.1]).tolist(), [0.0, 0.0, float(np.sin(yaw / 2)), float(np.cos(yaw / 2))]],
                           pick_pose=[rng.uniform(-1, 1, 3).tolist(), rng.uniform(-1, 1, 4).tolist()]))
    rplace.append(dict(name='x', extent=[0.1, 0.1, 0.1], scale=1.0, grasp_id=2, grasp_side=[['z+', 1]],
                       pick_pose=[[0.1, 0.2, 0.3], [0, 0, 0, 1]]))       # no place_pose: zeros and mobility_id = scene_id
    rd = dt.robot_data_json_to_pt(dict(container=rcont, placements=rplace, stats=dict(scene_id=77)), 'x')
    rec['robotjson/tray_dim'] = np.asarray(rcont['tray_dim'])
    rec['robotjson/tray_pose'] = np.asarray(rcont['tray_pose'])
    rec['robotjson/extent'] = np.asarray([p['extent'] for p in rplace])
    rec['robotjson/scale'] = np.asarray([p['scale'] for p in rplace])
    rec['robotjson/grasp_id'] = np.asarray([p['grasp_id'] for p in rplace], dtype=np.int64)
    rec['robotjson/grasp_side'] = np.asarray([[GRASP_KEYS.index(p['grasp_side'][0][0]), p['grasp_side'][0][1]] for p in rplace], dtype=np.int64)
    rec['robotjson/place_pos'] = np.asarray([p['place_pose'][0] for p in rplace[:5]])
    rec['robotjson/place_quat'] = np.asarray([p['place_pose'][1] for p in rplace[:5]])
    rec['robotjson/mobility'] = np.asarray([3000 + k for k in range(5)], dtype=np.int64)
    rec['robotjson/pick_pose'] = np.asarray([p['pick_pose'][0] + p['pick_pose'][1] for p in rplace], dtype=np.float64)
    rec['robotjson/ref_raw_x'] = rd.x.numpy()
    rec['robotjson/ref_raw_edges'] = np.asarray([[worlds.ROBOT_CONSTRAINTS.index(e[0]), e[1], e[2]] for e in rd.edge_index], dtype=np.int64)
    run('robotjson', rd.x.numpy().astype(np.float64), rd.edge_index, 'robot_box')
    # robot rows (29 columns): geometry 8 + the rest, world_dims from the container row
    rob = worlds.robot_box_batch(1, 4, seed=3)
    raw = np.concatenate([np.asarray([[0]] + [[1]] * 4, dtype=np.float64), rob.x.astype(np.float64)], axis=1)
    redges = [('gin', i, 0) for i in range(1, 5)] + [('gfree', j, i) for i in range(1, 4) for j in range(i + 1, 5)]
    run('robot', raw, redges, 'robot_box')
    np.savez_compressed(os.path.join(GOLD, 'pre_transform.npz'), **rec)
    print('pre_transform.npz')


def synth_weights(mode, H, seed):
    """untrained nn.Linear-style weights (single evaluations do not need contractive weights)"""
    dims = worlds.MODE_DIMS[mode]
    torch.manual_seed(seed)
    m = dfn.ConstraintDiffuser(dims=dims, hidden_dim=H, EBM='ULA', input_mode=mode, device='cpu', verbose=False)
    return {k: v.detach().numpy().astype(np.float32) for k, v in m.state_dict().items()}


def gen_single_eval_h256():
    """the BASELINE widths of C4 / C5: energy-mode gradients on 12-triangle graphs and the grasp branch on
    10-object robot_box graphs at hidden_dim 256"""
    gen_single_eval('single_eval_h256', 98, [
        ('t256e', 'diffuse_pairwise', 256, 'weights_diffuse_pairwise_h256_energy.npz', worlds.triangular_batch(3, 12, seed=27)),
        ('r256', 'robot_box', 256, 'weights_robot_box_h256.npz', worlds.robot_box_batch(3, 10, seed=28)),
    ])


def gen_single_eval_h128():
    """a hidden width between the two tuned ones (the reference's -hidden_dim is free, train_utils.py:107): direct and energy mode"""
    gen_single_eval('single_eval_h128', 95, [
        ('q128', 'qualitative', 128, 'weights_qualitative_h128.npz', worlds.qualitative_batch(3, 7, seed=61)),
        ('t128e', 'diffuse_pairwise', 128, 'weights_diffuse_pairwise_h128_energy.npz', worlds.triangular_batch(2, 12, seed=62)),
    ])


def gen_single_eval_box():
    """the reference's default dims (pose_dim 2, RandomSplitWorld boxes) at both widths"""
    gen_single_eval('single_eval_box', 97, [
        ('b64', 'diffuse_pairwise_box', 64, 'weights_diffuse_pairwise_box_h64.npz', worlds.box_batch(3, 6, seed=29)),
        ('b256', 'diffuse_pairwise_box', 256, 'weights_diffuse_pairwise_box_h256.npz', worlds.box_batch(2, 8, seed=30)),
    ])


def gen_single_eval(out_name='single_eval', rng_seed=99, cases=None):
    """single network evaluations, direct mode and energy mode"""
    rec = {}
    rng = np.random.default_rng(rng_seed)
    cases = cases or [
        ('q64', 'qualitative', 64, 'weights_qualitative_h64.npz', worlds.qualitative_batch(3, 8, seed=21)),
        ('q64small', 'qualitative', 64, 'weights_qualitative_h64.npz', worlds.qualitative_batch(1, 3, seed=22)),
        ('q256', 'qualitative', 256, 'weights_qualitative_h256.npz', worlds.qualitative_batch(2, 8, seed=23)),
        ('t64', 'diffuse_pairwise', 64, 'weights_diffuse_pairwise_h64.npz', worlds.triangular_batch(2, 12, seed=24)),
        ('t64e', 'diffuse_pairwise', 64, 'weights_diffuse_pairwise_h64_energy.npz', worlds.triangular_batch(2, 12, seed=25)),
        ('r64', 'robot_box', 64, 'weights_robot_box_h64.npz', worlds.robot_box_batch(2, 10, seed=26)),
    ]
    for tag, mode, H, wfile, batch in cases:
        W = oracle_mod.load_weights(os.path.join(GOLD, wfile))
        b = batch.to_torch()
        if tag == 'q64':
            # quirks: an edge whose type id matches no constraint is ignored (denoise_fn.py:512-517);
            # an isolated node divides 0/0 in the count normalisation (:523-524)
            ea = b.edge_attr.clone()
            ea[5] = 13.0
            ea[17] = 2.5
            b.edge_attr = ea
            keep = (b.edge_index != 3).all(dim=0)
            b.edge_index = b.edge_index[:, keep]
            b.edge_attr = b.edge_attr[keep]
        energy = tag.endswith('e')
        model, gd = build_reference(mode, H, W, energy=energy, EBM='MALA' if energy else 'ULA')
        P = worlds.MODE_DIMS[mode][-1][0]
        for key, val in batch_arrays(b).items():
            rec['%s/%s' % (tag, key)] = val
        ts = [0, 1, 37, 500, 998, 999]
        poses = (rng.standard_normal((len(ts), b.x.shape[0], P)) * 0.7).astype(np.float32)
        outs, grads, energies = [], [], []
        for i, t in enumerate(ts):
            tt = torch.tensor([t])
            if energy:
                g, e = model(torch.from_numpy(poses[i]).clone(), b, tt, eval=True, tag='EBM')
                grads.append(g.detach().numpy())
                energies.append(float(e.detach()))
            else:
                with torch.no_grad():
                    outs.append(model(torch.from_numpy(poses[i]).clone(), b, tt, eval=True).numpy())
        rec['%s/t' % tag] = np.asarray(ts, dtype=np.int32)
        rec['%s/poses' % tag] = poses
        if energy:
            rec['%s/grad' % tag] = np.stack(grads)
            rec['%s/energy' % tag] = np.asarray(energies, dtype=np.float64)
        else:
            rec['%s/out' % tag] = np.stack(outs)
        rec['%s/time_emb' % tag] = model.time_mlp(torch.tensor(ts)).detach().numpy()
    # untrained H=256 weights for every mode: regenerated from the seed by the reference here and
    # by the tests through the committed outputs only (weights are not stored: they are inputs of
    # a pure function of the seed -> stored as a checksum + outputs on the fixed graphs)
    np.savez_compressed(os.path.join(GOLD, out_name + '.npz'), **rec)
    print(out_name + '.npz')


def sd_batch(sizes, seed, shuffled=False):
    """qualitative graphs with the given numbers of tiles (<= 7: 8 tokens with the container)"""
    rng = np.random.default_rng(seed)
    gs = []
    for n in sizes:
        wd = worlds.sample_qualitative_world(rng, n)
        gs.append(worlds.encode_qualitative(wd['nodes'], wd['constraints']))
    b = worlds.collate(gs).to_torch()
    if shuffled:                                   # data.py-style per-graph permutation of the token positions
        sh = [torch.from_numpy(rng.permutation(n + 1)) for n in sizes]
        b.shuffled = torch.cat(sh).long()
    return b


def gen_struct_diffusion(H=64, out_name='struct_diffusion'):
    """single evaluations of the StructDiffusion baseline (denoise_fn.py:391-451): ragged graphs (3, 6, 8
    tokens -> padded and unpadded masks, head/graph mask mix-up with B = 3 and B = 2), batch.shuffled.
    H = 256 (round 6, struct_diffusion_h256.npz): the width bench.py --config sd runs (transformer width 512: the f16x2 GEMM path), weights
    trained 1500 steps with the reference's loss (oracle/ref_train.py qualitative 256 1500 --model StructDiffusion --quantise --batch 64)"""
    rec = {}
    rng = np.random.default_rng(123 if H == 64 else 124)
    wfile = 'weights_qualitative_h%d_sd.npz' % H
    W = oracle_mod.load_weights(os.path.join(GOLD, wfile))
    model, gd = build_reference('qualitative', H, W, model_name='StructDiffusion')
    cases = [('ragged3', sd_batch((2, 5, 7), 51)), ('full2', sd_batch((7, 7), 52)), ('single', sd_batch((3,), 53)),
             ('shuffled', sd_batch((4, 7, 2, 6), 54, shuffled=True))]
    if H != 64:
        cases.append(('ragged8', sd_batch((1, 7, 3, 6, 2, 5, 7, 4), 55)))
    for tag, b in cases:
        for key, val in batch_arrays(b).items():
            rec['%s/%s' % (tag, key)] = val
        rec['%s/batch' % tag] = b.batch.numpy().astype(np.int64)
        if hasattr(b, 'shuffled'):
            rec['%s/shuffled' % tag] = b.shuffled.numpy().astype(np.int64)
        ts = [0, 3, 500, 999]
        poses = (rng.standard_normal((len(ts), b.x.shape[0], 4)) * 0.7).astype(np.float32)
        outs = []
        for i, t in enumerate(ts):
            with torch.no_grad():
                outs.append(model(torch.from_numpy(poses[i]).clone(), b, torch.tensor([t]), eval=True).numpy())
        rec['%s/t' % tag] = np.asarray(ts, dtype=np.int32)
        rec['%s/poses' % tag] = poses
        rec['%s/out' % tag] = np.stack(outs)
    np.savez_compressed(os.path.join(GOLD, out_name + '.npz'), **rec)
    print(out_name + '.npz')


def gen_stability():
    """'stability_flat' (three constraint types, denoise_fn.py:209-210): the reference's data for it needs pybullet, its
    network does not -- single evaluations and a short ULA chain on a synthetic graph with all three edge types.
    The (untrained, seeded) weights travel inside the fixture."""
    mode, H = 'stability_flat', 64
    W = synth_weights(mode, H, 77)
    model, gd = build_reference(mode, H, W, T=50, S=3)
    bn = worlds.qualitative_batch(4, 5, seed=88)
    bn.edge_attr = (bn.edge_attr.astype(np.int64) % 3).astype(np.float32)
    b = bn.to_torch()
    rec = {'w/' + k: v for k, v in W.items()}
    rec.update(batch_arrays(b))
    rng = np.random.default_rng(5)
    ts = [0, 17, 49]
    poses = (rng.standard_normal((len(ts), b.x.shape[0], 4)) * 0.7).astype(np.float32)
    outs = []
    for i, t in enumerate(ts):
        with torch.no_grad():
            outs.append(model(torch.from_numpy(poses[i]).clone(), b, torch.tensor([t]), eval=True).numpy())
    rec.update(t=np.asarray(ts, dtype=np.int32), poses=poses, out=np.stack(outs))
    with PatchedNoise(9) as pn, contextlib.redirect_stdout(io.StringIO()):
        final, hist = gd.sample(b.clone(), return_history=True)
    rec.update(final=final.detach().numpy(), hist=np.stack([h.detach().numpy() for h in hist]), seed=np.int64(9), n_randn=np.int64(pn.c))
    np.savez_compressed(os.path.join(GOLD, 'stability.npz'), **rec)
    print('stability.npz  |final|max %.3g  |hist|max %.3g' % (np.abs(rec['final']).max(), np.abs(rec['hist']).max()))


def gen_robot_energy():
    """energy mode with a grasp group (robot_box, K_in = 6H): the reference's autograd gradient and batch energy on a synthetic
    graph, untrained seeded weights inside the fixture"""
    mode, H = 'robot_box', 64
    dims = worlds.MODE_DIMS[mode]
    torch.manual_seed(78)
    m = dfn.ConstraintDiffuser(dims=dims, hidden_dim=H, EBM='MALA', input_mode=mode, energy_wrapper=True, device='cpu', verbose=False)
    W = {k: v.detach().numpy().astype(np.float32) for k, v in m.state_dict().items()}
    b = worlds.robot_box_batch(2, 6, seed=79).to_torch()
    rec = {'w/' + k: v for k, v in W.items()}
    rec.update(batch_arrays(b))
    rng = np.random.default_rng(6)
    ts = [0, 400, 999]
    P = dims[-1][0]
    poses = (rng.standard_normal((len(ts), b.x.shape[0], P)) * 0.7).astype(np.float32)
    grads, energies = [], []
    for i, t in enumerate(ts):
        g, e = m(torch.from_numpy(poses[i]).clone(), b, torch.tensor([t]), eval=True, tag='EBM')
        grads.append(g.detach().numpy())
        energies.append(float(e.detach()))
    rec.update(t=np.asarray(ts, dtype=np.int32), poses=poses, grad=np.stack(grads), energy=np.asarray(energies, dtype=np.float64))
    np.savez_compressed(os.path.join(GOLD, 'robot_energy.npz'), **rec)
    print('robot_energy.npz  |grad|max %.3g  E %s' % (np.abs(rec['grad']).max(), energies))


def gen_options():
    """constructor options the other fixtures leave at their defaults: ConstraintDiffuser(normalize=False) (denoise_fn.py:523),
    GaussianDiffusion(step_sizes='0.5*self.betas', samples_per_step=<tensor schedule>, betas=<custom>) (ddpm.py:169-228)"""
    mode, H, T = 'qualitative', 64, 60
    W = oracle_mod.load_weights(os.path.join(GOLD, 'weights_qualitative_h64.npz'))
    dims = worlds.MODE_DIMS[mode]
    model = dfn.ConstraintDiffuser(dims=dims, hidden_dim=H, EBM='ULA', input_mode=mode, normalize=False, device='cpu', verbose=False)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    betas = np.linspace(1e-4, 0.05, T).astype(np.float64)
    sps = torch.tensor([1 + (t % 3) for t in range(T)])
    gd = ddpm.GaussianDiffusion(model, timesteps=T, EBM='ULA', betas=betas, samples_per_step=sps, step_sizes='0.5*self.betas').eval()
    b = worlds.qualitative_batch(2, 4, seed=91).to_torch()
    rec = dict(batch_arrays(b))
    rng = np.random.default_rng(7)
    poses = (rng.standard_normal((2, b.x.shape[0], 4)) * 0.7).astype(np.float32)
    with torch.no_grad():
        rec['out'] = np.stack([model(torch.from_numpy(poses[i]).clone(), b, torch.tensor([t]), eval=True).numpy() for i, t in enumerate((3, 55))])
    rec['poses'], rec['t'] = poses, np.asarray([3, 55], dtype=np.int32)
    with PatchedNoise(13) as pn, contextlib.redirect_stdout(io.StringIO()):
        final, hist = gd.sample(b.clone(), return_history=True)
    rec.update(final=final.detach().numpy(), hist=np.stack([h.detach().numpy() for h in hist]), seed=np.int64(13), n_randn=np.int64(pn.c),
               betas=betas, sps=sps.numpy().astype(np.int32), step_sizes=gd.step_sizes.numpy().astype(np.float32),
               posterior_log_variance_clipped=gd.posterior_log_variance_clipped.numpy())
    np.savez_compressed(os.path.join(GOLD, 'options.npz'), **rec)
    print('options.npz  randn calls %d  |final|max %.3g  |hist|max %.3g' % (pn.c, np.abs(rec['final']).max(), np.abs(rec['hist']).max()))


def build_composed_reference(H, W_robot, W_qual, weight=(1, 1), T=1000, S=10, EBM='ULA', energy=False, dtype=torch.float32):
    """the reference's composed denoiser the way its code expects to be assembled (nothing in the repository does the
    assembly: pose_encoder_2 & co. are None after the constructor, denoise_fn.py:287-291): a 'robot_qualitative'
    ConstraintDiffuser holding the robot domain's weights, the qualitative model's thirteen type MLPs appended to its
    ModuleList (so that type i >= 2 is qualitative type i - 2, denoise_fn.py:24,310-311), and the qualitative model's
    encoders / decoder / time MLP as the *_2 modules."""
    if dtype == torch.float64:
        torch.set_default_dtype(torch.float64)
    try:
        model = dfn.ConstraintDiffuser(dims=worlds.MODE_DIMS['robot_box'], hidden_dim=H, EBM=EBM, input_mode='robot_qualitative',
                                       device='cpu', verbose=False)
        model.load_state_dict({k: torch.from_numpy(v).to(dtype) for k, v in W_robot.items()})
        qual = dfn.ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=H, EBM=EBM, input_mode='qualitative',
                                      device='cpu', verbose=False)
        qual.load_state_dict({k: torch.from_numpy(v).to(dtype) for k, v in W_qual.items()})
        model.mlps.extend(qual.mlps)
        model.pose_encoder_2, model.geom_encoder_2 = qual.pose_encoder, qual.geom_encoder
        model.pose_decoder_2, model.time_mlp_2 = qual.pose_decoder, qual.time_mlp
        model.composing_weight = tuple(weight)
        if energy:      # the composed model in energy mode: its forward returns (dE/dposes, E) (denoise_fn.py:539-548), wrapped like any energy model
            model.energy_wrapper = True
        gd = ddpm.GaussianDiffusion(dfn.ComposedEBMDenoiseFn(model) if energy else model, timesteps=T, EBM=EBM, samples_per_step=S, step_sizes='2*self.betas')
        if dtype == torch.float64:      # (as build_reference: the custom kappa buffer is a plain attribute, step_sizes a tensor expression of the fp32 betas)
            gd = gd.double()
            gd._sqrt_recipm1_alphas_cumprod_custom = gd._sqrt_recipm1_alphas_cumprod_custom.double()
            gd.step_sizes = gd.step_sizes.double() if torch.is_tensor(gd.step_sizes) else gd.step_sizes
    finally:
        torch.set_default_dtype(torch.float32)
    return model, gd.eval()


def ref_single_timestep(gd, batch, x, t, seed, dtype=torch.float32):
    """ONE iteration of GaussianDiffusion.p_sample_loop's body from the state x, for the Metropolis samplers.  Every piece is the reference's
    own code (p_sample, AnnealedMALASampler / AnnealedMUHASampler.sample_step, the denoiser); only the glue around them is restated, line by
    line: the gradient / energy closures (ddpm.py:279-289), the sampler construction (:301-317), the loop body (:322-334), and the noise draws
    keep the call numbers they have in the full chain.  Used for ONE thing: the reference's own fp32-vs-fp64 disagreement over a single
    timestep (the noise floor a parity bar may be quoted against).  Checked where it is used: in fp32 it reproduces the recorded chain's
    successor BIT FOR BIT."""
    T, EBM = int(gd.num_timesteps), gd.EBM
    S = 4 if EBM == 'HMC' else int(gd.samples_per_step)
    per_t = 1 + S + (1 if EBM == 'HMC' else 0)
    eps = int(getattr(gd.denoise_fn, 'ebm_per_steps', 1))
    m = batch.mask
    gt_features = batch.x[:, gd.dims[-1][1]:gd.dims[-1][2]].to(dtype)
    shape = gt_features.shape

    def gradient_function(xx, bb, tt):
        return - gd.denoise_fn(xx, bb, tt, eval=True) * gd._sqrt_recipm1_alphas_cumprod_custom[tt]

    def energy_function(xx, bb, tt):
        return - gd.denoise_fn.neg_logp_unnorm(xx, bb, tt, eval=True) * gd._sqrt_recipm1_alphas_cumprod_custom[tt]

    def noise_function():
        return torch.randn(shape)
    if EBM == 'MALA':
        sampler = ddpm.AnnealedMALASampler(gd.samples_per_step, gd.step_sizes, gradient_function, noise_function, energy_function)
    elif EBM == 'ULA':                                             # (round 6; ddpm.py:291-300)
        sampler = ddpm.AnnealedULASampler(gd.samples_per_step, gd.step_sizes, gradient_function, noise_function)
    else:
        sampler = ddpm.AnnealedMUHASampler(4, gd.step_sizes, 0, 9 * gd.betas, 2, gradient_function=gradient_function, energy_function=energy_function)
    rates = []
    orig_upd = ddpm.MetropolisSampler._update_acceptance_rate
    ddpm.MetropolisSampler._update_acceptance_rate = lambda self, accept_rate, tt, debug=False: rates.append(float(accept_rate))
    pn = PatchedNoise(seed, dtype)
    # draws consumed before timestep t: the initial state, one per ancestral step, the sampler's own on the timesteps where it ran (j % eps == 0, ddpm.py:330)
    ran = sum(1 for j in range(t + 1, T) if j % eps == 0)
    pn.c, pn.uc = 1 + (T - 1 - t) + ran * (per_t - 1), ran * S
    b = batch.clone()
    b.x = b.x.to(dtype)
    prev = torch.is_grad_enabled()
    torch.set_grad_enabled(True)
    torch.set_default_dtype(dtype)
    try:
        with pn, contextlib.redirect_stdout(io.StringIO()):
            tt = torch.full((1,), t, dtype=torch.long)
            pose = gd.p_sample(b, torch.as_tensor(x).to(dtype).clone(), tt, tag='EBM')
            if t % eps == 0:
                pose = sampler.sample_step(pose, b, tt)
            pose[m.bool()] = gt_features[m.bool()].clone()
    finally:
        torch.set_default_dtype(torch.float32)
        torch.set_grad_enabled(prev)
        ddpm.MetropolisSampler._update_acceptance_rate = orig_upd
    return pose.detach().numpy(), (rates[-1] if rates else 0.0)


def gen_eps2_h256():
    """ebm_per_steps = 2 (ddpm.py:330; ComposedEBMDenoiseFn(model, 2), train_utils.py:284) at hidden_dim 256, energy-mode ULA S = 5, T = 200, three
    12-triangle graphs.  The chain passes a 2e22 transient and is CHAOTIC in the reference itself (its own fp32 and fp64 runs end 4e-3 apart and are
    unrelated at timestep 150), so the fixture records EVERY state and, per timestep, the reference's own fp64 successor of the recorded fp32 state
    (next_f64): an implementation is checked timestep by timestep from the recorded states, with a bar quoted against the reference's own
    fp32-vs-fp64 disagreement over that timestep.  The single-timestep glue is checked to reproduce the recorded fp32 chain bit for bit."""
    name, wfile, T, S, seed = 'chain_t256_ula_energy_eps2', 'weights_diffuse_pairwise_h256_energy.npz', 200, 5, 7
    batch = worlds.triangular_batch(3, 12, seed=73).to_torch()
    # ONE thread: with several, PyTorch-CPU's autograd of the energy (the scatter of the gather's backward) is not run-to-run deterministic at this
    # size -- two runs of the reference differ by ~1e-6 relative after one timestep (measured here: 1.5e-5 at |x| = 15), which this chain's gain
    # turns into unrelated states -- and the bit-for-bit check of the single-timestep glue below could not hold
    torch.set_num_threads(1)
    run_chain(name, 'diffuse_pairwise', 256, wfile, batch, 'ULA', T=T, S=S, seed=seed, energy=True, ebm_per_steps=2, full_hist=True)
    z = dict(np.load(os.path.join(GOLD, name + '.npz')))
    W = oracle_mod.load_weights(os.path.join(GOLD, wfile))
    _, gd32 = build_reference('diffuse_pairwise', 256, W, energy=True, EBM='ULA', T=T, S=S, ebm_per_steps=2)
    _, gd64 = build_reference('diffuse_pairwise', 256, W, energy=True, EBM='ULA', T=T, S=S, ebm_per_steps=2, dtype=torch.float64)
    nxt = []
    for k in range(T):
        x32, _ = ref_single_timestep(gd32, batch, z['hist'][k], T - 1 - k, seed)
        assert np.array_equal(x32, z['hist'][k + 1], equal_nan=True), ('single-timestep glue != chain', k)
        x64, _ = ref_single_timestep(gd64, batch, z['hist'][k].astype(np.float64), T - 1 - k, seed, torch.float64)
        nxt.append(x64)
    z['next_f64'] = np.stack(nxt).astype(np.float64)
    fl = [float(np.abs(nxt[k] - z['hist'][k + 1]).max() / (1.0 + np.abs(z['hist'][k + 1]).max())) for k in range(T)]
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **z)
    print('   fp32-vs-fp64 reference over ONE timestep from the same state: max %.2e, median %.2e, first ten: %s' %
          (max(fl), float(np.median(fl)), ' '.join('%.1e' % v for v in fl[:10])))


def gen_composed():
    """domain composition, input_mode 'robot_qualitative' (denoise_fn.py:287-291,310-311,341-371,487-503): the operator
    (_process_constraint on second-domain types), single evaluations with composing weights (1, 1) and (0.5, 2), and
    ULA chains, at hidden_dim 64 and 256"""
    rec = {}
    rng = np.random.default_rng(96)
    for tag, H, batch in (('c64', 64, worlds.robot_qualitative_batch(3, 6, seed=51)),
                          ('c256', 256, worlds.robot_qualitative_batch(2, 8, seed=52))):
        Wr = oracle_mod.load_weights(os.path.join(GOLD, 'weights_robot_box_h%d.npz' % H))
        Wq = oracle_mod.load_weights(os.path.join(GOLD, 'weights_qualitative_h%d.npz' % H))
        b = batch.to_torch()
        for key, val in batch_arrays(b).items():
            rec['%s/%s' % (tag, key)] = val
        ts = [0, 1, 37, 500, 998, 999]
        poses = (rng.standard_normal((len(ts), b.x.shape[0], 5)) * 0.7).astype(np.float32)
        rec['%s/t' % tag] = np.asarray(ts, dtype=np.int32)
        rec['%s/poses' % tag] = poses
        n = 9
        geoms_in = rng.uniform(0.05, 0.9, (n, 2, 2)).astype(np.float32)
        poses_in = rng.uniform(-1, 1, (n, 2, 4)).astype(np.float32)
        tval = np.asarray([417.25], dtype=np.float32)
        for wtag, weight in (('w11', (1, 1)), ('w052', (0.5, 2.0))):
            model, _ = build_composed_reference(H, Wr, Wq, weight)
            outs = []
            for i, t in enumerate(ts):
                with torch.no_grad():
                    outs.append(model(torch.from_numpy(poses[i]).clone(), b, torch.tensor([t]), eval=True).numpy())
            rec['%s/out_%s' % (tag, wtag)] = np.stack(outs)
            if wtag == 'w11':
                # energy mode of the composed model (denoise_fn.py:373-375,539-548 on the widened outputs): autograd gradients
                model.energy_wrapper = True
                grads, ens = [], []
                for i, t in enumerate(ts):
                    g, e = model(torch.from_numpy(poses[i]).clone(), b, torch.tensor([t]), eval=True, tag='EBM')
                    grads.append(g.detach().numpy())
                    ens.append(float(e.detach()))
                model.energy_wrapper = False
                rec['%s/grad' % tag] = np.stack(grads)
                rec['%s/energy' % tag] = np.asarray(ens, dtype=np.float64)
            # the operator on second-domain types: inputs built the way forward builds them (denoise_fn.py:497-503,322-334)
            with torch.no_grad():
                ge = model.geom_encoder_2(torch.from_numpy(geoms_in))
                pe = model.pose_encoder_2(torch.from_numpy(poses_in))
                te = model.time_mlp_2(torch.from_numpy(tval))
                d = {'args': None, 'geoms_emb_2': ge, 'poses_emb_2': pe, 'time_embedding': te.repeat(n, 1)}
                op = np.stack([model._process_constraint(i, d).numpy() for i in range(2, len(model.mlps))])
            rec.update({'%s/op_geoms_in' % tag: geoms_in, '%s/op_poses_in' % tag: poses_in, '%s/op_t' % tag: tval,
                        '%s/op_out_%s' % (tag, wtag): op})
    np.savez_compressed(os.path.join(GOLD, 'composed.npz'), **rec)
    print('composed.npz')
    # chains
    for name, H, batch, T, S, weight, seed in (('chain_c64_ula', 64, worlds.robot_qualitative_batch(2, 6, seed=53), 1000, 3, (1, 1), 11),
                                               ('chain_c256_ula', 256, worlds.robot_qualitative_batch(2, 6, seed=54), 200, 3, (0.5, 0.5), 12)):
        Wr = oracle_mod.load_weights(os.path.join(GOLD, 'weights_robot_box_h%d.npz' % H))
        Wq = oracle_mod.load_weights(os.path.join(GOLD, 'weights_qualitative_h%d.npz' % H))
        model, gd = build_composed_reference(H, Wr, Wq, weight, T=T, S=S)
        b = batch.to_torch()
        t0 = time.time()
        with PatchedNoise(seed) as pn, contextlib.redirect_stdout(io.StringIO()):
            out, hist = gd.sample(b.clone(), return_history=True)
        dt = time.time() - t0
        out = out.detach().numpy()
        hist = np.stack([h.detach().numpy() for h in hist])
        idx = sorted(set(i for i in HIST_IDX if i <= T) | {T})
        r = dict(batch_arrays(b))
        r.update(final=out, hist_idx=np.asarray(idx, dtype=np.int32), hist=hist[idx], seed=np.int64(seed), T=np.int32(T), S=np.int32(S),
                 H=np.int32(H), n_randn=np.int64(pn.c), weight=np.asarray(weight, dtype=np.float32), ref_seconds=np.float64(dt))
        np.savez_compressed(os.path.join(GOLD, name + '.npz'), **r)
        print('%-20s %6.1fs  randn calls %d  |final|max %.3f  |hist|max %.3g' % (name, dt, pn.c, np.abs(out).max(), np.abs(hist).max()), flush=True)


def gen_composed_energy_chains():
    """ULA on the ENERGY gradient of a composed model (ComposedEBMDenoiseFn around the 'robot_qualitative' ConstraintDiffuser with
    energy_wrapper=True: ddpm.py:940-966 with the epsilon of denoise_fn.py:539-548), every timestep recorded"""
    for name, H, batch, T, S, seed in (('chain_c64_ula_energy', 64, worlds.robot_qualitative_batch(2, 6, seed=55), 60, 2, 13),
                                       ('chain_c256_ula_energy', 256, worlds.robot_qualitative_batch(2, 6, seed=56), 40, 2, 14)):
        # hidden_dim 64: both domains' weights trained in energy mode (oracle/ref_train.py --energy); 256: the direct-mode fixtures.
        # Either way the REFERENCE chain overflows within a few timesteps: the zero column's energy term cnt * p_z^2 alone has ULA
        # gain |1 - 2 beta kappa 2 cnt| >> 1 at beta -> 0.999 -- the fixture pins the evaluations along that transient and the
        # non-finite end state
        sfx = '_energy' if H == 64 else ''
        Wr = oracle_mod.load_weights(os.path.join(GOLD, 'weights_robot_box_h%d%s.npz' % (H, sfx)))
        Wq = oracle_mod.load_weights(os.path.join(GOLD, 'weights_qualitative_h%d%s.npz' % (H, sfx)))
        model, gd = build_composed_reference(H, Wr, Wq, (1, 1), T=T, S=S, energy=True)
        b = batch.to_torch()
        t0 = time.time()
        with PatchedNoise(seed) as pn, contextlib.redirect_stdout(io.StringIO()):
            out, hist = gd.sample(b.clone(), return_history=True)
        dt = time.time() - t0
        out = out.detach().numpy()
        hist = np.stack([h.detach().numpy() for h in hist])
        r = dict(batch_arrays(b))
        r.update(final=out, hist_idx=np.arange(T + 1, dtype=np.int32), hist=hist, seed=np.int64(seed), T=np.int32(T), S=np.int32(S),
                 H=np.int32(H), n_randn=np.int64(pn.c), weight=np.asarray((1, 1), dtype=np.float32), ref_seconds=np.float64(dt))
        np.savez_compressed(os.path.join(GOLD, name + '.npz'), **r)
        print('%-24s %6.1fs  randn calls %d  |final|max %.3g  |hist|max %.3g' % (name, dt, pn.c, np.abs(out).max(), np.abs(hist).max()), flush=True)


def gen_composed_metropolis(which=()):
    """MALA and HMC of the REFERENCE on a composed energy model (AnnealedMALASampler / AnnealedMUHASampler, ddpm.py:999-1047,1050-1128, over
    ComposedEBMDenoiseFn around the 'robot_qualitative' ConstraintDiffuser with energy_wrapper=True; denoise_fn.py:287-291,310-311,341-371,
    487-503), the reference's default schedule (step_sizes '2*self.betas'), EVERY state recorded together with the reference's own
    acceptance log (MetropolisSampler._update_acceptance_rate), like chain_t256_mala.  Unlike ULA on the same energy (chain_c*_ula_energy:
    overflows fp32 within a dozen timesteps) the Metropolis chains stay finite: the first timesteps climb to 1e14 through the ancestral
    step's gain on the zero column, the accept test rejects what would grow further, and from t ~ 950 on the T = 1000 MALA chain sits at
    |x| ~ 1 with mixed acceptance.  HMC at T = 1000 never accepts (its leapfrog uses the step size and mass of index 0..3, ddpm.py:1076-1084),
    so its fixtures use T = 20 / T = 8 like chain_t64_hmc_T20."""
    jobs = (('chain_c64_mala', 64, 'MALA', 1000, 2, 13, 55), ('chain_c256_mala', 256, 'MALA', 200, 2, 14, 56),
            ('chain_c64_hmc', 64, 'HMC', 20, 4, 15, 57), ('chain_c256_hmc', 256, 'HMC', 8, 4, 16, 58), ('chain_c256_hmc_T20', 256, 'HMC', 20, 4, 17, 59))
    for name, H, EBM, T, S, seed, bseed in jobs:
        if which and name not in which:
            continue
        sfx = '_energy' if H == 64 else ''          # hidden_dim 64: both domains trained in energy mode; 256: the direct-mode fixtures (as chain_c256_ula_energy)
        Wr = oracle_mod.load_weights(os.path.join(GOLD, 'weights_robot_box_h%d%s.npz' % (H, sfx)))
        Wq = oracle_mod.load_weights(os.path.join(GOLD, 'weights_qualitative_h%d%s.npz' % (H, sfx)))
        model, gd = build_composed_reference(H, Wr, Wq, (1, 1), T=T, S=S, EBM=EBM, energy=True)
        b = worlds.robot_qualitative_batch(2, 6, seed=bseed).to_torch()
        rates = {}
        orig_upd = ddpm.MetropolisSampler._update_acceptance_rate

        def record(self, accept_rate, t, debug=False):
            rates[int(t)] = float(accept_rate)
            return orig_upd(self, accept_rate, t, debug)
        ddpm.MetropolisSampler._update_acceptance_rate = record
        t0 = time.time()
        try:
            with PatchedNoise(seed) as pn, contextlib.redirect_stdout(io.StringIO()):
                out, hist = gd.sample(b.clone(), return_history=True)
        finally:
            ddpm.MetropolisSampler._update_acceptance_rate = orig_upd
        dt = time.time() - t0
        out = out.detach().numpy()
        hist = np.stack([h.detach().numpy() for h in hist])
        r = dict(batch_arrays(b))
        if EBM == 'HMC':
            # the reference's OWN sensitivity, timestep by timestep: the fp64 reference run for one timestep from each recorded fp32 state.  The leapfrog
            # map of this energy amplifies rounding differences of the gradient evaluations several thousand-fold at hidden_dim 256; a parity bar
            # on the successor state is quoted against THIS disagreement, not against anything an implementation under test measured.
            _, gd32 = build_composed_reference(H, Wr, Wq, (1, 1), T=T, S=S, EBM=EBM, energy=True)
            _, gd64 = build_composed_reference(H, Wr, Wq, (1, 1), T=T, S=S, EBM=EBM, energy=True, dtype=torch.float64)
            nxt64, acc64 = [], []
            for k in range(T):
                x32, a32 = ref_single_timestep(gd32, b, hist[k], T - 1 - k, seed)
                assert np.array_equal(x32, hist[k + 1], equal_nan=True) and abs(a32 - rates.get(T - 1 - k, 0.0)) < 1e-9, ('single-timestep glue != chain', name, k)
                x64, a64 = ref_single_timestep(gd64, b, hist[k].astype(np.float64), T - 1 - k, seed, torch.float64)
                nxt64.append(x64)
                acc64.append(a64)
            r.update(next_f64=np.stack(nxt64).astype(np.float64), accept_f64=np.asarray(acc64, dtype=np.float64))
            fl = [float(np.abs(nxt64[k] - hist[k + 1]).max() / (1.0 + np.abs(hist[k + 1]).max())) for k in range(T)]
            print('   fp32-vs-fp64 reference, one timestep from the same state: max relative difference %.2e (per timestep: %s); acceptance equal: %s'
                  % (max(fl), ' '.join('%.1e' % v for v in fl), bool(np.allclose(acc64, r.get('accept', [rates.get(t, 0.0) for t in range(T - 1, -1, -1)]) if False else [rates.get(T - 1 - k, 0.0) for k in range(T)]))))
        r.update(final=out, hist_idx=np.arange(T + 1, dtype=np.int32), hist=hist, seed=np.int64(seed), T=np.int32(T), S=np.int32(S),
                 S_accept=np.int32(4 if EBM == 'HMC' else S), H=np.int32(H), n_randn=np.int64(pn.c), n_rand=np.int64(pn.uc),
                 weight=np.asarray((1, 1), dtype=np.float32), accept=np.asarray([rates.get(t, 0.0) for t in range(T)], dtype=np.float32),
                 ref_seconds=np.float64(dt), sampler=np.asarray(EBM))
        np.savez_compressed(os.path.join(GOLD, name + '.npz'), **r)
        print('%-20s %6.1fs  randn %d rand %d  finite %s  |hist|max %.3g  |final|max %.3g  acceptance mean %.3f  distinct rates %d' %
              (name, dt, pn.c, pn.uc, bool(np.isfinite(hist).all()), np.nanmax(np.abs(hist)), np.nanmax(np.abs(out)), float(r['accept'].mean()),
               len(set(np.round(r['accept'], 4)))), flush=True)


def gen_chains(which):
    jobs = {
        'chain_sd64_ula': lambda: run_chain('chain_sd64_ula', 'qualitative', 64, 'weights_qualitative_h64_sd.npz',
                                            sd_batch((3, 7, 5), 61), 'ULA', S=4, model_name='StructDiffusion'),
        'chain_sd64_noebm': lambda: run_chain('chain_sd64_noebm', 'qualitative', 64, 'weights_qualitative_h64_sd.npz',
                                              sd_batch((6, 2), 62), False, model_name='StructDiffusion'),
        # round 6: what was oracle-only at the benchmark width -- the transformer baseline at hidden_dim 256 (width 512: the
        # f16x2 GEMM k_sd_gemm_h2) under the bench's sampler setting, ULA+ and ebm_per_steps = 2 (ddpm.py:297-299,330) at hidden_dim 256
        'chain_sd256_ula': lambda: run_chain('chain_sd256_ula', 'qualitative', 256, 'weights_qualitative_h256_sd.npz',
                                             sd_batch((3, 7, 5, 6), 71), 'ULA', S=10, model_name='StructDiffusion'),
        # the reference's own fp32-vs-fp64 disagreement along that chain: the first timesteps' Langevin steps have gain > 1, so its history
        # checkpoints are compared with a bar derived from this (tests: max(2e-3, 8 x the reference's own disagreement))
        'chain_sd256_ula_f64': lambda: run_chain('chain_sd256_ula_f64', 'qualitative', 256, 'weights_qualitative_h256_sd.npz',
                                                 sd_batch((3, 7, 5, 6), 71), 'ULA', S=10, model_name='StructDiffusion', dtype=torch.float64),
        'chain_q256_ulaplus': lambda: run_chain('chain_q256_ulaplus', 'qualitative', 256, 'weights_qualitative_h256.npz',
                                                worlds.qualitative_batch(2, 6, seed=72).to_torch(), 'ULA+'),
        'chain_t256_mala_eps2': lambda: run_chain('chain_t256_mala_eps2', 'diffuse_pairwise', 256, 'weights_diffuse_pairwise_h256_energy.npz',
                                                  worlds.triangular_batch(2, 12, seed=74).to_torch(), 'MALA', T=100, S=4, energy=True,
                                                  ebm_per_steps=2, full_hist=True),
        'chain_q64_T1000_B4': lambda: run_chain('chain_q64_T1000_B4', 'qualitative', 64, 'weights_qualitative_h64.npz',
                                                worlds.qualitative_batch(4, 8, seed=31).to_torch(), 'ULA'),
        'chain_q64_T100_B1': lambda: run_chain('chain_q64_T100_B1', 'qualitative', 64, 'weights_qualitative_h64.npz',
                                               worlds.qualitative_batch(1, 3, seed=5).to_torch(), 'ULA', T=100),
        'chain_q256_T100_B1': lambda: run_chain('chain_q256_T100_B1', 'qualitative', 256, 'weights_qualitative_h256.npz',
                                                worlds.qualitative_batch(1, 3, seed=5).to_torch(), 'ULA', T=100),
        'chain_q256_T1000_B4': lambda: run_chain('chain_q256_T1000_B4', 'qualitative', 256, 'weights_qualitative_h256.npz',
                                                 worlds.qualitative_batch(4, 8, seed=32).to_torch(), 'ULA'),
        'chain_q64_noebm': lambda: run_chain('chain_q64_noebm', 'qualitative', 64, 'weights_qualitative_h64.npz',
                                             mixed_batch(), False),
        'chain_q64_ulaplus': lambda: run_chain('chain_q64_ulaplus', 'qualitative', 64, 'weights_qualitative_h64.npz',
                                               worlds.qualitative_batch(1, 5, seed=33).to_torch(), 'ULA+'),
        'chain_t64_mala': lambda: run_chain('chain_t64_mala', 'diffuse_pairwise', 64, 'weights_diffuse_pairwise_h64_energy.npz',
                                            worlds.triangular_batch(2, 12, seed=34).to_torch(), 'MALA', S=2, energy=True),
        'chain_t64_hmc': lambda: run_chain('chain_t64_hmc', 'diffuse_pairwise', 64, 'weights_diffuse_pairwise_h64_energy.npz',
                                           worlds.triangular_batch(2, 12, seed=37).to_torch(), 'HMC', energy=True),
        'chain_t64_hmc_T20': lambda: run_chain('chain_t64_hmc_T20', 'diffuse_pairwise', 64, 'weights_diffuse_pairwise_h64_energy.npz',
                                               worlds.triangular_batch(2, 12, seed=38).to_torch(), 'HMC', T=20, energy=True),
        'chain_t64_ula_energy_eps2': lambda: run_chain('chain_t64_ula_energy_eps2', 'diffuse_pairwise', 64, 'weights_diffuse_pairwise_h64_energy.npz',
                                                       worlds.triangular_batch(2, 8, seed=39).to_torch(), 'ULA', T=100, S=3, energy=True,
                                                       ebm_per_steps=2),
        'chain_t64_ula': lambda: run_chain('chain_t64_ula', 'diffuse_pairwise', 64, 'weights_diffuse_pairwise_h64.npz',
                                           worlds.triangular_batch(2, 12, seed=35).to_torch(), 'ULA', S=3),
        # the reference's default dims ((2,0,2),(2,2,4)): pose_dim 2 (RandomSplitWorld boxes, input_mode 'diffuse_pairwise')
        'chain_b64_ula': lambda: run_chain('chain_b64_ula', 'diffuse_pairwise_box', 64, 'weights_diffuse_pairwise_box_h64.npz',
                                           worlds.box_batch(3, 6, seed=46).to_torch(), 'ULA', S=5),
        'chain_b256_ula': lambda: run_chain('chain_b256_ula', 'diffuse_pairwise_box', 256, 'weights_diffuse_pairwise_box_h256.npz',
                                            worlds.box_batch(2, 8, seed=47).to_torch(), 'ULA', T=200, S=10),
        'chain_q128_ula': lambda: run_chain('chain_q128_ula', 'qualitative', 128, 'weights_qualitative_h128.npz',
                                            worlds.qualitative_batch(3, 6, seed=63).to_torch(), 'ULA', T=200, S=5),
        'chain_t128_mala': lambda: run_chain('chain_t128_mala', 'diffuse_pairwise', 128, 'weights_diffuse_pairwise_h128_energy.npz',
                                             worlds.triangular_batch(2, 12, seed=64).to_torch(), 'MALA', T=100, S=2, energy=True, full_hist=True),
        'chain_r64_ula': lambda: run_chain('chain_r64_ula', 'robot_box', 64, 'weights_robot_box_h64.npz',
                                           worlds.robot_box_batch(2, 10, seed=36).to_torch(), 'ULA', S=5),
        # BASELINE configs C4 / C5 at their hidden width (H = 256): 12-triangle graphs under MALA S = 10 with the
        # reference's logged acceptance per timestep and EVERY state recorded; 10-object robot_box graphs under ULA S = 10
        'chain_t256_mala': lambda: run_chain('chain_t256_mala', 'diffuse_pairwise', 256, 'weights_diffuse_pairwise_h256_energy.npz',
                                             worlds.triangular_batch(4, 12, seed=44).to_torch(), 'MALA', S=10, energy=True, full_hist=True),
        'chain_r256_ula': lambda: run_chain('chain_r256_ula', 'robot_box', 256, 'weights_robot_box_h256.npz',
                                            worlds.robot_box_batch(3, 10, seed=45).to_torch(), 'ULA', S=10),
        # HMC at the BASELINE hidden width (round 5): T = 20 where its proposals are accepted and rejected (every state + the acceptance log),
        # and T = 100 (sparser acceptance)
        'chain_t256_hmc_T20': lambda: run_chain('chain_t256_hmc_T20', 'diffuse_pairwise', 256, 'weights_diffuse_pairwise_h256_energy.npz',
                                                worlds.triangular_batch(3, 12, seed=48).to_torch(), 'HMC', T=20, energy=True),
        'chain_t256_hmc_T100': lambda: run_chain('chain_t256_hmc_T100', 'diffuse_pairwise', 256, 'weights_diffuse_pairwise_h256_energy.npz',
                                                 worlds.triangular_batch(2, 12, seed=49).to_torch(), 'HMC', T=100, energy=True, full_hist=True),
        # bench.py's C2 weights on 8-object graphs: what the REFERENCE sampler does with them (finite rows, non-finite rows,
        # final poses for the solved check) -- the HIP path must show the same rows and the same solved mask
        'chain_q256_bench_B16': lambda: run_chain('chain_q256_bench_B16', 'qualitative', 256, 'weights/qualitative_h256_ref30k.npz',
                                                  worlds.qualitative_batch(16, 8, seed=19).to_torch(), 'ULA'),
        # (round 1-4 kept a 50 000-step checkpoint of the earlier recipe for the MIXED case -- some graphs overflow, some do not; round 5 gets
        # that case from the 300 000-step reference-recipe weights under the reference's documented S = 3, below, and dropped the 10 MB file)
        # the reference recipe AS WRITTEN (30 000 fixed worlds of 2-5 objects, 300 000 Adam steps, tools/train_gpu.py TRAIN_RECIPE=reference;
        # profiles/r03_train_reference_recipe_log.txt): what the REFERENCE sampler does with the final weights on 8-object graphs
        'chain_q256_ref300k_B16': lambda: run_chain('chain_q256_ref300k_B16', 'qualitative', 256, 'weights/qualitative_h256_ref300k.npz',
                                                    worlds.qualitative_batch(16, 8, seed=19).to_torch(), 'ULA'),
        # ... and the same weights under samples_per_step = 3 (the reference's documented command line, train_ddpm.py:31-35): a MIXED batch -- most
        # graphs come back from the transient, some overflow fp32 and stay non-finite
        'chain_q256_ref300k_S3_B32': lambda: run_chain('chain_q256_ref300k_S3_B32', 'qualitative', 256, 'weights/qualitative_h256_ref300k.npz',
                                                       worlds.qualitative_batch(32, 8, seed=21).to_torch(), 'ULA', S=3),
        'chain_q64_T1000_B4_f64': lambda: run_chain('chain_q64_T1000_B4_f64', 'qualitative', 64, 'weights_qualitative_h64.npz',
                                                    worlds.qualitative_batch(4, 8, seed=31).to_torch(), 'ULA', dtype=torch.float64),
    }
    for k, fn in jobs.items():
        if not which or k in which:
            fn()


def mixed_batch():
    rng = np.random.default_rng(41)
    gs = []
    for n in (2, 5, 8):
        wd = worlds.sample_qualitative_world(rng, n)
        gs.append(worlds.encode_qualitative(wd['nodes'], wd['constraints']))
    return worlds.collate(gs).to_torch()


if __name__ == '__main__':
    which = set(sys.argv[1:])
    os.makedirs(GOLD, exist_ok=True)
    if not which or 'schedule' in which:
        gen_schedule()
    if not which or 'operators' in which:
        gen_operators()
    if not which or 'evaluate_summary' in which:
        gen_evaluate_summary()
    if not which or 'labeller' in which:
        gen_labeller()
    if not which or 'single_eval' in which:
        gen_single_eval()
    if not which or 'single_eval_h256' in which:
        gen_single_eval_h256()
    if not which or 'single_eval_box' in which:
        gen_single_eval_box()
    if not which or 'single_eval_h128' in which:
        gen_single_eval_h128()
    if not which or 'composed' in which:
        gen_composed()
    if not which or 'composed_energy' in which:
        gen_composed_energy_chains()
    if not which or 'composed_metropolis' in which or any(w.startswith('chain_c') and ('mala' in w or 'hmc' in w) for w in which):
        gen_composed_metropolis(tuple(w for w in which if w.startswith('chain_c')))
    if not which or 'pre_transform' in which:
        gen_pre_transform()
    if not which or 'struct_diffusion' in which:
        gen_struct_diffusion()
    if not which or 'struct_diffusion_h256' in which:
        gen_struct_diffusion(256, 'struct_diffusion_h256')
    if not which or 'stability' in which:
        gen_stability()
    if not which or 'robot_energy' in which:
        gen_robot_energy()
    if not which or 'options' in which:
        gen_options()
    if not which or 'chain_t256_ula_energy_eps2' in which:
        gen_eps2_h256()
    gen_chains(which)
