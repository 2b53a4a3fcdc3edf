"""Domain composition (the reference's input_mode 'robot_qualitative': networks/denoise_fn.py:287-291,310-311,341-371,
487-503): a robot_box model and a qualitative model evaluated on ONE graph that carries both kinds of constraints.

`-m "not gpu"`: the CPU restatement (oracle/compose.py) against vectors the reference itself produced
(oracle/gen_golden.py:gen_composed -> tests/golden/composed.npz, chain_c64_ula.npz, chain_c256_ula.npz).
`-m gpu`: the HIP path (ccsp_compose_denoise / ccsp_compose_chain_run behind ConstraintDiffuser.compose) against the same
vectors and against the restatement on fresh inputs.  Tolerances as for the single-domain tests: 2e-5 relative on single
evaluations, 1e-4 absolute on a chain's final poses."""
import numpy as np
import pytest
import torch

import compose as compose_oracle
from conftest import golden, golden_batch, oracle_model, rel_err, weights, worlds
from diffusion_ccsp_amd import noise

CASES = [('c64', 64), ('c256', 256)]
WEIGHTS = {'w11': (1, 1), 'w052': (0.5, 2.0)}


def _oracle_pair(H, T=1000, S=10):
    return (oracle_model('robot_box', H, 'weights_robot_box_h%d.npz' % H, T=T, S=S),
            oracle_model('qualitative', H, 'weights_qualitative_h%d.npz' % H, T=T, S=S))


@pytest.mark.parametrize('tag,H', CASES)
def test_oracle_single_evaluation_vs_reference(tag, H):
    z = golden('composed')
    m1, m2 = _oracle_pair(H)
    b = golden_batch(z, tag + '/')
    assert (z[tag + '/edge_attr'] >= 2).any() and (z[tag + '/edge_attr'] < 2).any()       # both domains present
    for wtag, w in WEIGHTS.items():
        g = compose_oracle.ComposedOracleGraph(m1, m2, b, weight=w)
        for i, t in enumerate(z[tag + '/t']):
            want = z['%s/out_%s' % (tag, wtag)][i]
            got = g.denoise(z[tag + '/poses'][i], int(t))
            assert rel_err(got, want) < 2e-5, (tag, wtag, int(t))


@pytest.mark.parametrize('name,H', [('chain_c64_ula', 64), ('chain_c256_ula', 256)])
def test_oracle_chain_vs_reference(name, H):
    z = golden(name)
    T, S = int(z['T']), int(z['S'])
    m1, m2 = _oracle_pair(H, T=T, S=S)
    b = golden_batch(z)
    g = compose_oracle.ComposedOracleGraph(m1, m2, b, weight=tuple(z['weight']))
    N = z['x'].shape[0]
    zs = noise.normal_stream(int(z['seed']), int(z['n_randn']), N, 5)
    final, hist = g.chain(zs, S, history=True)
    assert np.abs(final - z['final']).max() < 1e-4
    for k, idx in enumerate(z['hist_idx']):
        assert rel_err(hist[idx], z['hist'][k]) < 2e-3, (name, int(idx))


def _energy_pair(H, T, S):
    sfx = '_energy' if H == 64 else ''
    return (oracle_model('robot_box', H, 'weights_robot_box_h%d%s.npz' % (H, sfx), T=T, S=S, energy=True),
            oracle_model('qualitative', H, 'weights_qualitative_h%d%s.npz' % (H, sfx), T=T, S=S, energy=True))


def _energy_chain_errors(step, z):
    """ULA on the ENERGY gradient of a composed model (chain_c{64,256}_ula_energy: the reference's ComposedEBMDenoiseFn around the
    energy_wrapper 'robot_qualitative' model).  The reference's own chain overflows fp32 within a dozen timesteps (the zero column's
    energy term alone has ULA gain >> 1 at beta -> 0.999), so every timestep is run from the reference's RECORDED state: while the
    recorded successor is finite it must be reproduced to 1e-4 relative, afterwards the same rows must be non-finite"""
    T = int(z['T'])
    bad, checked = [], 0
    for k in range(T):
        if not np.isfinite(z['hist'][k]).all():
            break
        got, want = step(z['hist'][k], T - 1 - k), z['hist'][k + 1]
        fin = np.isfinite(want).all(axis=1)
        if not np.array_equal(np.isfinite(got).all(axis=1), fin) or (fin.any() and rel_err(got[fin], want[fin]) > 1e-4):
            bad.append((k, rel_err(got[fin], want[fin]) if fin.any() else None))
        checked += 1
    assert checked >= 8
    return bad


@pytest.mark.parametrize('name,H', [('chain_c64_ula_energy', 64), ('chain_c256_ula_energy', 256)])
def test_oracle_energy_chain_vs_reference(name, H):
    z = golden(name)
    T, S = int(z['T']), int(z['S'])
    m1, m2 = _energy_pair(H, T, S)
    g = compose_oracle.ComposedOracleGraph(m1, m2, golden_batch(z), weight=(1, 1))
    zs = noise.normal_stream(int(z['seed']), int(z['n_randn']), z['x'].shape[0], 5)
    with np.errstate(all='ignore'):
        bad = _energy_chain_errors(lambda x, t: g.chain(zs, S, energy=True, x=x, t_first=t, t_last=t), z)
    assert not bad, bad


@pytest.mark.parametrize('tag,H', CASES)
def test_oracle_energy_mode_vs_reference(tag, H):
    """energy and autograd gradients of the reference's composed model (tag='EBM', energy_wrapper) at composing_weight (1, 1)"""
    z = golden('composed')
    m1 = oracle_model('robot_box', H, 'weights_robot_box_h%d.npz' % H, energy=True)
    m2 = oracle_model('qualitative', H, 'weights_qualitative_h%d.npz' % H, energy=True)
    g = compose_oracle.ComposedOracleGraph(m1, m2, golden_batch(z, tag + '/'))
    for i, t in enumerate(z[tag + '/t']):
        grad, E = g.energy_grad(z[tag + '/poses'][i], int(t))
        assert abs(E - z[tag + '/energy'][i]) <= 2e-5 * (1 + abs(z[tag + '/energy'][i])), int(t)
        assert rel_err(grad, z[tag + '/grad'][i]) < 5e-5, int(t)
        assert np.abs(z[tag + '/grad'][i][:, 2]).max() > 0          # the zero column's own term


# ---------------------------------------------------------------------------------------------------------------- GPU
def _composed_model(H, device, weight=(1, 1), T=1000, S=10, EBM='ULA'):
    from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion
    first = ConstraintDiffuser(dims=worlds.MODE_DIMS['robot_box'], hidden_dim=H, input_mode='robot_qualitative', device=device, verbose=False)
    first.load_state_dict({k: torch.from_numpy(v) for k, v in weights('weights_robot_box_h%d.npz' % H).items()})
    second = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=H, input_mode='qualitative', device=device, verbose=False)
    second.load_state_dict({k: torch.from_numpy(v) for k, v in weights('weights_qualitative_h%d.npz' % H).items()})
    first.compose(second, composing_weight=weight)
    gd = GaussianDiffusion(first, timesteps=T, EBM=EBM, samples_per_step=S)
    return first, second, gd


@pytest.mark.gpu
@pytest.mark.parametrize('tag,H', CASES)
def test_hip_single_evaluation_vs_reference(tag, H, device):
    z = golden('composed')
    b = golden_batch(z, tag + '/')
    for wtag, w in WEIGHTS.items():
        model, second, _ = _composed_model(H, device, w)
        assert len(model.constraint_sets) == 15 and model.pose_encoder_2 is not None and model.composing_weight == tuple(w)
        for i, t in enumerate(z[tag + '/t']):
            got = model(torch.from_numpy(z[tag + '/poses'][i]), b, torch.tensor([int(t)])).cpu().numpy()
            assert rel_err(got, z['%s/out_%s' % (tag, wtag)][i]) < 2e-5, (tag, wtag, int(t))
        # the operator on second-domain types, inputs built from the *_2 encoders like forward does (denoise_fn.py:497-503)
        ge = model.geom_encoder_2(torch.from_numpy(z[tag + '/op_geoms_in']))
        pe = model.pose_encoder_2(torch.from_numpy(z[tag + '/op_poses_in']))
        te = model.time_mlp_2(torch.from_numpy(z[tag + '/op_t']))
        n = pe.shape[0]
        d = {'args': None, 'geoms_emb_2': ge, 'poses_emb_2': pe, 'time_embedding': te.repeat(n, 1)}
        want = z['%s/op_out_%s' % (tag, wtag)]
        for i in range(2, 15):
            got = model._process_constraint(i, d).cpu().numpy()
            assert got.shape == (n, 2, 5) and np.all(got[:, :, 2] == 0)
            assert rel_err(got, want[i - 2]) < 2e-5, (tag, wtag, i)
        # energy mode (tag='EBM' with energy_wrapper): built for composing_weight (1, 1); other weights must refuse, not fall back
        model.energy_wrapper = True
        model._drop_handle()
        if tuple(w) == (1, 1):
            for i, t in enumerate(z[tag + '/t']):
                grad, E = model(torch.from_numpy(z[tag + '/poses'][i]), b, torch.tensor([int(t)]), tag='EBM')
                assert abs(float(E) - z[tag + '/energy'][i]) <= 2e-5 * (1 + abs(z[tag + '/energy'][i])), (tag, int(t))
                assert rel_err(grad.cpu().numpy(), z[tag + '/grad'][i]) < 5e-5, (tag, int(t))
        else:
            with pytest.raises(NotImplementedError):
                model(torch.from_numpy(z[tag + '/poses'][0]), b, torch.tensor([3]), tag='EBM')
        model.energy_wrapper = False
        model._drop_handle()


@pytest.mark.gpu
@pytest.mark.parametrize('name,H', [('chain_c64_ula', 64), ('chain_c256_ula', 256)])
def test_hip_chain_vs_reference(name, H, device):
    z = golden(name)
    T, S = int(z['T']), int(z['S'])
    model, second, gd = _composed_model(H, device, tuple(float(v) for v in z['weight']), T=T, S=S)
    b = golden_batch(z)
    x, hist = gd.p_sample_loop(b, return_history=True, seed=int(z['seed']))
    x = x.cpu().numpy()
    assert np.abs(x - z['final']).max() < 1e-4
    for k, idx in enumerate(z['hist_idx']):
        assert rel_err(hist[int(idx)].cpu().numpy(), z['hist'][k]) < 2e-3, (name, int(idx))
    # a chain split across calls reproduces the unsplit chain bit for bit (noise draws are indexed by call number)
    mid = T // 2
    x1 = gd.p_sample_segment(b, hist[T - mid], mid - 1, 0, seed=int(z['seed']))
    assert np.array_equal(x1.cpu().numpy(), x)


@pytest.mark.gpu
@pytest.mark.parametrize('name,H', [('chain_c64_ula_energy', 64), ('chain_c256_ula_energy', 256)])
def test_hip_energy_chain_vs_reference(name, H, device):
    """ccsp_compose_chain_run on two energy_wrapper models: every evaluation is the composed energy gradient; per-timestep parity
    from the reference's recorded states (see _energy_chain_errors); the direct output of the same models (forward with
    tag != 'EBM', denoise_fn.py:535-537) is still available"""
    from diffusion_ccsp_amd import ComposedEBMDenoiseFn, ConstraintDiffuser, GaussianDiffusion
    z = golden(name)
    T, S = int(z['T']), int(z['S'])
    sfx = '_energy' if H == 64 else ''
    first = ConstraintDiffuser(dims=worlds.MODE_DIMS['robot_box'], hidden_dim=H, input_mode='robot_qualitative', EBM='ULA', energy_wrapper=True,
                               device=device, verbose=False)
    first.load_state_dict(weights('weights_robot_box_h%d%s.npz' % (H, sfx)))
    second = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=H, input_mode='qualitative', EBM='ULA', energy_wrapper=True,
                                device=device, verbose=False)
    second.load_state_dict(weights('weights_qualitative_h%d%s.npz' % (H, sfx)))
    first.compose(second, (1, 1))
    gd = GaussianDiffusion(ComposedEBMDenoiseFn(first), timesteps=T, EBM='ULA', samples_per_step=S)
    b = golden_batch(z)
    seed = int(z['seed'])
    bad = _energy_chain_errors(lambda x, t: gd.p_sample_segment(b, torch.from_numpy(x), t, t, seed=seed).cpu().numpy(), z)
    assert not bad, bad
    # the whole chain ends where the reference's ends: every row non-finite or equal
    x = gd.sample(b, seed=seed).cpu().numpy()
    assert np.array_equal(np.isfinite(x).all(axis=1), np.isfinite(z['final']).all(axis=1))
    # direct output of the energy_wrapper pair = the direct composed evaluation of the same weights
    poses = z['hist'][0]
    d_energy = first(torch.from_numpy(poses), b, torch.tensor([7]), tag='none').cpu().numpy()
    first.energy_wrapper = False
    first._drop_handle()
    d_direct = first(torch.from_numpy(poses), b, torch.tensor([7])).cpu().numpy()
    first.energy_wrapper = True
    first._drop_handle()
    assert np.array_equal(d_energy, d_direct)
    # (MALA and HMC on the pair: test_hip_composed_mala_vs_oracle / _hmc_)


def _composed_mala_oracle(H, T, S, b, seed, N):
    m1, m2 = _energy_pair(H, T, S)
    g = compose_oracle.ComposedOracleGraph(m1, m2, b, weight=(1, 1))
    zs = noise.normal_stream(seed, 1 + T * (1 + S), N, 5)
    us = noise.uniform_stream(seed, T * S, N)
    with np.errstate(all='ignore'):
        x, hist = g.chain(zs, S, sampler='MALA', energy=True, history=True, uniform=us)
    return g, zs, us, x, hist


def test_oracle_composed_mala_chain_is_a_metropolis_chain():
    """MALA on the composed energy (oracle/compose.py; no reference golden: the composed energy and gradient it is built on are pinned
    by `composed.npz`, the accept step restates ddpm.py:1013-1041): every inner step leaves a node row either where it was or at the
    proposal, conditioned rows stay fixed, acceptance rates are rates"""
    z = golden('chain_c64_ula_energy')
    b = golden_batch(z)
    T, S, seed = 6, 3, 11
    g, zs, us, x, hist = _composed_mala_oracle(64, T, S, b, seed, z['x'].shape[0])
    assert hist.shape[0] == T + 1 and len(g.last_accept) == T * S and all(0.0 <= a <= 1.0 for a in g.last_accept)
    m = z['mask'].astype(bool)
    p0 = g.m1.dims[-1][1]
    assert m.any() and np.array_equal(hist[-1][m], g.x[m][:, p0:p0 + 5])


@pytest.mark.gpu
@pytest.mark.parametrize('name,H', [('chain_c64_ula_energy', 64), ('chain_c256_ula_energy', 256)])
def test_hip_composed_mala_vs_oracle(name, H, device):
    """ccsp_compose_chain_run with CCSP_SAMPLER_MALA on two energy_wrapper models (the reference's MALA on its composed model: ddpm.py:999-1047
    over gradient_function / energy_function :280-289 of the 'robot_qualitative' ConstraintDiffuser): every timestep from the ORACLE's
    recorded state (the composed energy chain leaves fp32 range within a dozen timesteps, see _energy_chain_errors) -- finite successors
    within 1e-4 relative, the same rows non-finite otherwise, at most one flipped near-tie accept; acceptance rates reported"""
    from diffusion_ccsp_amd import ComposedEBMDenoiseFn, ConstraintDiffuser, GaussianDiffusion
    z = golden(name)
    b = golden_batch(z)
    T, S, seed = 10, 3, 11
    sfx = '_energy' if H == 64 else ''
    g, zs, us, x_o, hist = _composed_mala_oracle(H, T, S, b, seed, z['x'].shape[0])
    first = ConstraintDiffuser(dims=worlds.MODE_DIMS['robot_box'], hidden_dim=H, input_mode='robot_qualitative', EBM='MALA', energy_wrapper=True,
                               device=device, verbose=False)
    first.load_state_dict(weights('weights_robot_box_h%d%s.npz' % (H, sfx)))
    second = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=H, input_mode='qualitative', EBM='MALA', energy_wrapper=True,
                                device=device, verbose=False)
    second.load_state_dict(weights('weights_qualitative_h%d%s.npz' % (H, sfx)))
    first.compose(second, (1, 1))
    gd = GaussianDiffusion(ComposedEBMDenoiseFn(first), timesteps=T, EBM='MALA', samples_per_step=S)
    bad, checked = [], 0
    for k in range(T):
        if not np.isfinite(hist[k]).all():
            break
        got = gd.p_sample_segment(b, torch.from_numpy(hist[k]), T - 1 - k, T - 1 - k, seed=seed).cpu().numpy()
        want = hist[k + 1]
        fin = np.isfinite(want).all(axis=1)
        if not np.array_equal(np.isfinite(got).all(axis=1), fin) or (fin.any() and rel_err(got[fin], want[fin]) > 1e-4):
            bad.append((k, rel_err(got[fin], want[fin]) if fin.any() else None))
        checked += 1
    assert checked >= 3 and len(bad) <= 1, (checked, bad)
    rates = gd.last_accept_rates.cpu().numpy()
    assert rates.shape == (T,) and (rates >= 0).all() and (rates <= 1).all()
    x = gd.sample(b, seed=seed).cpu().numpy()
    m = z['mask'].astype(bool)
    p0 = g.m1.dims[-1][1]
    assert np.array_equal(x[m], g.x[m][:, p0:p0 + 5])
    # where proposals ARE accepted: late timesteps of the T = 1000 schedule from a small state (mixed acceptance, 0.14 .. 1.0 per inner step)
    T, tf, n_t = 1000, 200, 6
    m1, m2 = _energy_pair(H, T, S)
    g = compose_oracle.ComposedOracleGraph(m1, m2, b, weight=(1, 1))
    N = z['x'].shape[0]
    zs = noise.normal_stream(seed, 1 + T * (1 + S), N, 5)
    us = noise.uniform_stream(seed, T * S, N)
    x0 = (0.3 * np.random.RandomState(0).randn(N, 5)).astype(np.float32)
    with np.errstate(all='ignore'):
        _, hist = g.chain(zs, S, sampler='MALA', energy=True, history=True, uniform=us, x=x0, t_first=tf, t_last=tf - n_t + 1)
    acc_o = np.asarray(g.last_accept).reshape(n_t, S).mean(axis=1)
    assert 0.2 < acc_o.mean() < 1.0 and np.isfinite(hist).all()
    gd = GaussianDiffusion(ComposedEBMDenoiseFn(first), timesteps=T, EBM='MALA', samples_per_step=S)
    flips = 0
    for k in range(n_t):
        t = tf - k
        got = gd.p_sample_segment(b, torch.from_numpy(hist[k]), t, t, seed=seed).cpu().numpy()
        rate = float(gd.last_accept_rates[t].cpu())
        if rel_err(got, hist[k + 1]) > 1e-4 or abs(rate - acc_o[k]) > 1e-6:
            flips += 1                                           # a near-tie accept decided the other way moves one node row by one proposal
            assert abs(rate - acc_o[k]) <= 1.0 / (N * S) + 1e-6 and rel_err(got, hist[k + 1]) < 0.2, (t, rate, acc_o[k])
    assert flips <= 1


@pytest.mark.gpu
def test_hip_fresh_inputs_vs_restatement(device):
    """inputs the fixtures do not hold: ragged graphs, a graph with no second-domain edge, unmatched type ids, no normalisation"""
    H = 64
    m1, m2 = _oracle_pair(H)
    b = worlds.robot_qualitative_batch(3, 5, seed=77).to_torch()
    ea = b.edge_attr.clone()
    g0 = (b.edge_index[0] < 6)
    keep = ~(g0 & (ea >= 2))                                   # graph 0: robot constraints only
    b.edge_index, ea = b.edge_index[:, keep], ea[keep]
    ea[3] = 15.0                                               # beyond both domains: ignored (denoise_fn.py:512-517)
    b.edge_attr = ea
    rng = np.random.default_rng(5)
    poses = (rng.standard_normal((b.x.shape[0], 5)) * 0.7).astype(np.float32)
    for normalize in (True, False):
        model, second, _ = _composed_model(H, device, (1, 0.25))
        model.normalize = normalize
        g = compose_oracle.ComposedOracleGraph(m1, m2, b, weight=(1, 0.25), normalize=normalize)
        for t in (0, 400, 999):
            got = model(torch.from_numpy(poses), b, torch.tensor([t])).cpu().numpy()
            assert rel_err(got, g.denoise(poses, t)) < 2e-5, (normalize, t)


@pytest.mark.gpu
def test_hip_degenerate_domains_and_fresh_chain(device):
    """a batch with NO second-domain edge, one with NO first-domain edge (an empty sub-graph on one side of the composition), and a
    short ULA chain on a ragged batch against the numpy restatement (seeded noise, T = 40, S = 2)"""
    H = 64
    rng = np.random.default_rng(9)
    for keep_first, keep_second in ((True, False), (False, True)):
        m1, m2 = _oracle_pair(H)
        b = worlds.robot_qualitative_batch(2, 4, seed=81).to_torch()
        sel = (b.edge_attr < 2) if keep_first else (b.edge_attr >= 2)
        b.edge_index, b.edge_attr = b.edge_index[:, sel].contiguous(), b.edge_attr[sel].contiguous()
        poses = (rng.standard_normal((b.x.shape[0], 5)) * 0.7).astype(np.float32)
        model, second, _ = _composed_model(H, device)
        g = compose_oracle.ComposedOracleGraph(m1, m2, b)
        got = model(torch.from_numpy(poses), b, torch.tensor([321])).cpu().numpy()
        want = g.denoise(poses, 321)
        assert np.array_equal(np.isnan(got), np.isnan(want))          # nodes without any edge: 0 / 0 like the reference
        ok = ~np.isnan(want)
        assert rel_err(got[ok], want[ok]) < 2e-5, (keep_first, keep_second)
    # a short chain on ragged graphs
    T, S = 40, 2
    m1, m2 = _oracle_pair(H, T=T, S=S)
    gs = []
    for n_obj, seed in ((3, 1), (6, 2), (4, 3)):
        one = worlds.robot_qualitative_batch(1, n_obj, seed=90 + seed)
        gs.append(dict(x=one.x, edge_index=one.edge_index, edge_attr=one.edge_attr, mask=one.mask, world_dims=one.world_dims[0]))
    b = worlds.collate(gs).to_torch()
    model, second, gd = _composed_model(H, device, (1, 0.5), T=T, S=S)
    x, hist = gd.p_sample_loop(b, return_history=True, seed=77)
    zs = noise.normal_stream(77, gd.n_normal_calls(), b.x.shape[0], 5)
    want, whist = compose_oracle.ComposedOracleGraph(m1, m2, b, weight=(1, 0.5)).chain(zs, S, history=True)
    assert np.abs(x.cpu().numpy() - want).max() < 1e-4 * (1 + np.abs(want).max())
    for k in (1, 5, 20, T):
        assert rel_err(hist[k].cpu().numpy(), whist[k]) < 2e-3, k


def _composed_pair_hip(H, device, EBM):
    from diffusion_ccsp_amd import ConstraintDiffuser
    sfx = '_energy' if H == 64 else ''
    first = ConstraintDiffuser(dims=worlds.MODE_DIMS['robot_box'], hidden_dim=H, input_mode='robot_qualitative', EBM=EBM, energy_wrapper=True,
                               device=device, verbose=False)
    first.load_state_dict(weights('weights_robot_box_h%d%s.npz' % (H, sfx)))
    second = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=H, input_mode='qualitative', EBM=EBM, energy_wrapper=True,
                                device=device, verbose=False)
    second.load_state_dict(weights('weights_qualitative_h%d%s.npz' % (H, sfx)))
    first.compose(second, (1, 1))
    return first


def _hmc_oracle_segment(H, b, N, seed, T, tf, n_t, x0):
    m1, m2 = _energy_pair(H, T, 4)
    g = compose_oracle.ComposedOracleGraph(m1, m2, b, weight=(1, 1))
    zs = noise.normal_stream(seed, 1 + T * 6, N, 5)
    us = noise.uniform_stream(seed, T * 4, N)
    with np.errstate(all='ignore'):
        _, hist = g.chain(zs, 4, sampler='HMC', energy=True, history=True, uniform=us, x=x0, t_first=tf, t_last=tf - n_t + 1)
    return g, hist


def test_oracle_composed_hmc_chain_is_a_metropolis_chain():
    """HMC on the composed energy (oracle/compose.py restating ddpm.py:1087-1128, 917-937 on the composed gradient / energy): rates are
    rates, conditioned rows stay fixed, and with these step sizes (inner indices 0..3 of the schedule) proposals are accepted"""
    z = golden('chain_c64_ula_energy')
    b = golden_batch(z)
    N = z['x'].shape[0]
    x0 = (0.3 * np.random.RandomState(0).randn(N, 5)).astype(np.float32)
    # (at T = 1000 the reference's HMC accepts nothing -- its leapfrog pairs the momentum scale of timestep t with the mass of inner index
    # 0..3, chain_t64_hmc.npz: acceptance 0 everywhere -- so the accept path is exercised on short schedules)
    g, hist = _hmc_oracle_segment(64, b, N, 11, 4, 3, 4, x0)
    acc = np.asarray(g.last_accept)
    assert acc.shape == (16,) and (acc >= 0).all() and (acc <= 1).all() and 0.5 < acc.mean() < 1.0 and np.isfinite(hist).all()
    m = z['mask'].astype(bool)
    p0 = g.m1.dims[-1][1]
    assert np.array_equal(hist[-1][m], g.x[m][:, p0:p0 + 5])


@pytest.mark.gpu
@pytest.mark.parametrize('name,H', [('chain_c64_ula_energy', 64), ('chain_c256_ula_energy', 256)])
def test_hip_composed_hmc_vs_oracle(name, H, device):
    """ccsp_compose_chain_run with CCSP_SAMPLER_HMC on two energy_wrapper models (ddpm.py:1050-1128 over the composed model's
    gradient_function / energy_function): every timestep of a T = 8 and a T = 4 schedule (where the reference's HMC accepts: rejected first
    refreshments, accepted later ones, mixed acceptance at T = 4), each from the oracle's recorded state -- acceptance rates equal, at most
    one flipped near-tie accept per schedule, successors within 1e-4 relative at hidden_dim 64.  At hidden_dim 256 the bar is 5e-2: the
    leapfrog map of THIS energy is explosive (the zero column's term has Hessian 2 cnt, and ss_i / mass_i^2 = 6e2 with kappa_i ss_i / 2 = 6e-3
    per half step: two leapfrogs amplify a gradient difference ~ 6e3-fold; the states grow tenfold per timestep), and the two
    implementations' single gradient evaluations differ by 1e-7 .. 9e-7 there (5e-8 .. 1e-7 at hidden_dim 64) -- measured: 1e-3 .. 2e-2 per
    timestep whatever the GEMM scheme (f16x2, bf16x3, fp32 MFMA), with every acceptance rate exact"""
    from diffusion_ccsp_amd import ComposedEBMDenoiseFn, GaussianDiffusion
    z = golden(name)
    b = golden_batch(z)
    N, seed = z['x'].shape[0], 11
    x0 = (0.3 * np.random.RandomState(0).randn(N, 5)).astype(np.float32)
    first = _composed_pair_hip(H, device, 'HMC')
    for T in (8, 4):
        g, hist = _hmc_oracle_segment(H, b, N, seed, T, T - 1, T, x0)
        acc_o = np.asarray(g.last_accept).reshape(T, 4).mean(axis=1)
        assert 0.3 < acc_o.mean() < 1.0 and np.isfinite(hist).all()
        gd = GaussianDiffusion(ComposedEBMDenoiseFn(first), timesteps=T, EBM='HMC', samples_per_step=4)
        flips = 0
        for k in range(T):
            t = T - 1 - k
            got = gd.p_sample_segment(b, torch.from_numpy(hist[k]), t, t, seed=seed).cpu().numpy()
            rate = float(gd.last_accept_rates[t].cpu())
            if rel_err(got, hist[k + 1]) > (1e-4 if H == 64 else 5e-2) or abs(rate - acc_o[k]) > 1e-6:
                flips += 1
                assert abs(rate - acc_o[k]) <= 1.0 / (N * 4) + 1e-6 and rel_err(got, hist[k + 1]) < 0.2, (T, t, rate, acc_o[k])
        assert flips <= 1, T


# ---- round 5: MALA and HMC on the composed energy model against the REFERENCE's own chains -------------------------------------------------
# chain_c{64,256}_mala / chain_c{64,256}_hmc / chain_c256_hmc_T20 (oracle/gen_golden.py gen_composed_metropolis): the imported reference running
# AnnealedMALASampler / AnnealedMUHASampler over ComposedEBMDenoiseFn(robot_qualitative energy model) on its default schedule, every state
# and its own acceptance log recorded.  Both the numpy restatement (oracle/compose.py) and the HIP path are started from every recorded
# state and must reproduce the next one (1e-4 of the state's magnitude) and the reference's mean acceptance of the timestep exactly.
REF_METROPOLIS = [('chain_c64_mala', 64, 'MALA'), ('chain_c256_mala', 256, 'MALA'), ('chain_c64_hmc', 64, 'HMC'), ('chain_c256_hmc', 256, 'HMC'),
                  ('chain_c256_hmc_T20', 256, 'HMC')]


def _ref_metropolis_fixture(name):
    z = golden(name)
    T = int(z['T'])
    assert z['hist'].shape[0] == T + 1 and z['accept'].shape == (T,) and str(z['sampler']) in ('MALA', 'HMC')
    return z


@pytest.mark.parametrize('name,H,sampler', REF_METROPOLIS)
def test_oracle_composed_metropolis_vs_reference(name, H, sampler):
    """oracle/compose.py's MALA / HMC on the composed energy against the reference's recorded chains (every 25th timestep of the
    T = 1000 chain, every 8th of the T = 200 one, every timestep of the short HMC chains)"""
    from test_oracle_golden import mala_timestep_errors
    z = _ref_metropolis_fixture(name)
    T, S, seed = int(z['T']), int(z['S']), int(z['seed'])
    b = golden_batch(z)
    N = z['x'].shape[0]
    m1, m2 = _energy_pair(H, T, S)
    g = compose_oracle.ComposedOracleGraph(m1, m2, b, weight=(1, 1))
    zs = noise.normal_stream(seed, int(z['n_randn']), N, 5)
    us = noise.uniform_stream(seed, int(z['n_rand']), N)
    n_acc = int(z['S_accept'])

    def step(x, t):
        with np.errstate(all='ignore'):
            _, hist = g.chain(zs, S, sampler=sampler, energy=True, history=True, uniform=us, x=x, t_first=t, t_last=t)
        acc = np.zeros(T, dtype=np.float64)
        acc[t] = float(np.mean(g.last_accept[-n_acc:]))
        return hist[-1], acc
    ts = list(range(T - 1, -1, -(25 if T == 1000 else (8 if T == 200 else 1))))
    bad = mala_timestep_errors(step, z, ts)
    assert len(bad) <= max(1, len(ts) // 50), bad
    if sampler == 'MALA':
        assert len(set(np.round(z['accept'], 3))) >= 4 and 0.0 < float(z['accept'].mean()) < 1.0       # the reference chain accepts AND rejects
    assert np.isfinite(z['hist']).all()


@pytest.mark.gpu
@pytest.mark.parametrize('name,H,sampler', REF_METROPOLIS)
def test_hip_composed_metropolis_vs_reference(name, H, sampler, device):
    """ccsp_compose_chain_run with CCSP_SAMPLER_MALA / _HMC against the reference's recorded chains: EVERY timestep from the reference's
    state.  A timestep may differ only where the accept kernel itself reports a near-tie (ccsp_chain_margins: |log acceptance ratio - log u|
    within fp32 rounding of its terms), and at most one timestep in a hundred may."""
    from diffusion_ccsp_amd import ComposedEBMDenoiseFn, GaussianDiffusion
    from test_hip_parity import NEAR_TIE
    from test_oracle_golden import mala_timestep_errors
    z = _ref_metropolis_fixture(name)
    T, S, seed = int(z['T']), int(z['S']), int(z['seed'])
    b = golden_batch(z)
    first = _composed_pair_hip(H, device, sampler)
    gd = GaussianDiffusion(ComposedEBMDenoiseFn(first), timesteps=T, EBM=sampler, samples_per_step=S)
    assert gd.n_normal_calls() == int(z['n_randn'])
    gd.record_margins = True

    def step(x, t):
        out = gd.p_sample_segment(b, torch.from_numpy(x), t, t, seed=seed).cpu().numpy()
        return out, gd.last_accept_rates.cpu().numpy(), gd.last_margins.cpu().numpy()
    bad = mala_timestep_errors(step, z, list(range(T - 1, -1, -1)))
    print(name, 'flagged timesteps (t, rows off, acceptance here, reference, smallest |margin| / scale):', bad)
    assert len(bad) <= max(1, T // 100), bad
    assert all(r[4] < NEAR_TIE for r in bad), bad


@pytest.mark.gpu
def test_hip_composed_mala_honours_the_shard_energy_hook(device):
    """a composed MALA chain must reduce its batch energies across shards like ccsp_chain_run does (the hook / communicator of the FIRST
    domain's model: sharding.enable_global_batch_energy), or the shards of a global batch decouple silently.  A one-rank "reduction" that leaves the
    pair alone reproduces the unhooked chain bit for bit and is called once per inner step; one that scales it changes the chain; composed HMC,
    which does not reduce, refuses a model with a hook installed."""
    from diffusion_ccsp_amd import CcspError, ComposedEBMDenoiseFn, GaussianDiffusion, sharding
    z = golden('chain_c64_mala')
    b = golden_batch(z)
    T, S, seed = int(z['T']), int(z['S']), int(z['seed'])
    t0 = 500                                                    # the reference accepts 0.3 .. 0.96 of the proposals at timesteps 500 .. 491
    first = _composed_pair_hip(64, device, 'MALA')
    gd = GaussianDiffusion(ComposedEBMDenoiseFn(first), timesteps=T, EBM='MALA', samples_per_step=S)
    x0 = torch.from_numpy(z['hist'][T - 1 - t0])

    class OneRank(object):
        def __init__(self, scale):
            self.scale, self.calls = scale, 0

        def all_reduce(self, t):
            t.mul_(self.scale)
            self.calls += 1
    plain = gd.p_sample_segment(b, x0, t0, t0 - 9, seed=seed).cpu().numpy()
    rates = gd.last_accept_rates.cpu().numpy()[t0 - 9:t0 + 1]
    assert 0.2 < rates.mean() < 0.9
    out = {}
    for scale in (1.0, 4.0):
        hk = OneRank(scale)
        sharding.enable_global_batch_energy(gd, hk)
        out[scale] = gd.p_sample_segment(b, x0, t0, t0 - 9, seed=seed).cpu().numpy()
        assert hk.calls == 10 * S
    assert np.array_equal(out[1.0], plain, equal_nan=True)
    assert not np.array_equal(out[4.0], plain, equal_nan=True)
    gd_h = GaussianDiffusion(ComposedEBMDenoiseFn(first), timesteps=T, EBM='HMC', samples_per_step=4)
    sharding.enable_global_batch_energy(gd_h, OneRank(1.0))
    with pytest.raises(CcspError, match='HMC chain does not reduce'):
        gd_h.p_sample_segment(b, x0, 9, 8, seed=seed)
    sharding.enable_global_batch_energy(gd_h, None)
    assert np.isfinite(gd_h.p_sample_segment(b, x0, 9, 8, seed=seed).cpu().numpy()).all()
