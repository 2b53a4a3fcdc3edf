"""The CPU oracle (oracle/ccsp_oracle.c) against golden vectors produced by the reference itself
(oracle/gen_golden.py).  This is what pins the checker; the reference has no tests of its own.
Tolerances: schedule buffers bit-exact; single evaluations 2e-5 relative (fp32 summation-order
noise of a 1280-long dot product); final poses of full chains 1e-4 (the north-star bound)."""
import numpy as np
import pytest
import torch

from conftest import (MODE_TYPES, golden, golden_batch, golden_meta, oracle, oracle_model, rel_err,
                      weights, worlds)


def test_schedule_bit_exact():
    z = golden('schedule')
    for T in (100, 1000):
        m = oracle.OracleModel(weights('weights_qualitative_h64.npz'), worlds.MODE_DIMS['qualitative'], 64, 13, timesteps=T)
        s = m.schedule()
        for k in ['betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_recip_alphas_cumprod',
                  'sqrt_recipm1_alphas_cumprod', 'posterior_log_variance_clipped', 'posterior_mean_coef1',
                  'posterior_mean_coef2', 'posterior_variance', 'kappa', 'step_sizes']:
            assert np.array_equal(s[k], z['T%d/%s' % (T, k)]), (T, k)
    # SURVEY 8a-1 probe values
    z1000 = {k: z['T1000/' + k] for k in ('betas', 'alphas_cumprod', 'kappa', 'step_sizes')}
    assert abs(z1000['betas'][0] - 4.128e-5) < 1e-7 and z1000['betas'][999] == np.float32(0.999)
    assert abs(z1000['kappa'][0] - 155.6) < 0.1 and abs(z1000['step_sizes'][999] - 1.998) < 1e-6


CASES = [('q64', 'qualitative', 64, 'weights_qualitative_h64.npz'),
         ('q64small', 'qualitative', 64, 'weights_qualitative_h64.npz'),
         ('q256', 'qualitative', 256, 'weights_qualitative_h256.npz'),
         ('t64', 'diffuse_pairwise', 64, 'weights_diffuse_pairwise_h64.npz'),
         ('r64', 'robot_box', 64, 'weights_robot_box_h64.npz')]


@pytest.mark.parametrize('tag,mode,H,wfile', CASES)
def test_single_evaluation_direct(tag, mode, H, wfile):
    z = golden('single_eval')
    m = oracle_model(mode, H, wfile)
    g = m.graph(golden_batch(z, tag + '/'))
    for i, t in enumerate(z[tag + '/t']):
        want = z[tag + '/out'][i]
        got = g.denoise(z[tag + '/poses'][i], int(t))
        assert np.array_equal(np.isnan(got), np.isnan(want))      # isolated node -> NaN row, like the reference
        ok = ~np.isnan(want)
        assert rel_err(got[ok], want[ok]) < 2e-5, (tag, t)
        assert rel_err(m.time_embedding(int(t)), z[tag + '/time_emb'][i]) < 1e-5
    if tag == 'q64':
        assert np.isnan(z[tag + '/out']).any()


BOX_CASES = [('b64', 'diffuse_pairwise_box', 64, 'weights_diffuse_pairwise_box_h64.npz'),
             ('b256', 'diffuse_pairwise_box', 256, 'weights_diffuse_pairwise_box_h256.npz')]


@pytest.mark.parametrize('tag,mode,H,wfile', BOX_CASES)
def test_single_evaluation_default_dims(tag, mode, H, wfile):
    """the reference's DEFAULT dims ((2,0,2),(2,2,4)) (denoise_fn.py:185, train_utils.py:273: RandomSplitWorld boxes, pose_dim 2)"""
    z = golden('single_eval_box')
    m = oracle_model(mode, H, wfile)
    g = m.graph(golden_batch(z, tag + '/'))
    assert z[tag + '/poses'].shape[-1] == 2
    for i, t in enumerate(z[tag + '/t']):
        assert rel_err(g.denoise(z[tag + '/poses'][i], int(t)), z[tag + '/out'][i]) < 2e-5, (tag, t)


def test_single_evaluation_h128():
    """a hidden width other than the two tuned ones (the reference's -hidden_dim is free, train_utils.py:107)"""
    z = golden('single_eval_h128')
    g = oracle_model('qualitative', 128, 'weights_qualitative_h128.npz').graph(golden_batch(z, 'q128/'))
    for i, t in enumerate(z['q128/t']):
        assert rel_err(g.denoise(z['q128/poses'][i], int(t)), z['q128/out'][i]) < 2e-5, int(t)
    g = oracle_model('diffuse_pairwise', 128, 'weights_diffuse_pairwise_h128_energy.npz', energy=True).graph(golden_batch(z, 't128e/'))
    for i, t in enumerate(z['t128e/t']):
        grad, E = g.energy_grad(z['t128e/poses'][i], int(t))
        assert abs(E - z['t128e/energy'][i]) <= 2e-5 * (1 + abs(z['t128e/energy'][i]))
        assert rel_err(grad, z['t128e/grad'][i]) < 5e-5, int(t)


def test_single_evaluation_energy():
    z = golden('single_eval')
    tag = 't64e'
    m = oracle_model('diffuse_pairwise', 64, 'weights_diffuse_pairwise_h64_energy.npz', energy=True)
    g = m.graph(golden_batch(z, tag + '/'))
    for i, t in enumerate(z[tag + '/t']):
        grad, E = g.energy_grad(z[tag + '/poses'][i], int(t))
        assert abs(E - z[tag + '/energy'][i]) <= 2e-5 * (1 + abs(z[tag + '/energy'][i]))
        assert rel_err(grad, z[tag + '/grad'][i]) < 5e-5, t


CHAINS = ['chain_q64_T1000_B4', 'chain_q64_T100_B1', 'chain_q256_T100_B1', 'chain_q64_noebm',
          'chain_q64_ulaplus', 'chain_t64_ula', 'chain_r64_ula', 'chain_b64_ula', 'chain_b256_ula', 'chain_q128_ula']


def _run_oracle_chain(z, f64=False, history=True):
    meta = golden_meta(z)
    EBM = {'False': False}.get(meta['EBM'], meta['EBM'])
    m = oracle_model(meta['mode'], int(z['H']), meta['weights'], T=int(z['T']), S=int(z['S']), energy=meta['energy'], f64=f64,
                     model=meta.get('model', 'Diffusion-CCSP'))
    g = m.graph(golden_batch(z))
    return g.chain(EBM, seed=int(z['seed']), history=history)


@pytest.mark.parametrize('name', CHAINS)
def test_full_chain_final_poses(name):
    z = golden(name)
    final, hist = _run_oracle_chain(z)
    # the north-star bound: final poses within 1e-4 of the reference sampler on identical noise
    assert np.abs(final - z['final']).max() < 1e-4, name
    assert np.abs(z['final']).max() < 50.0          # trained-like weights: the chain is contractive
    # history checkpoints (relative: early timesteps pass through very large transients)
    for k, idx in enumerate(z['hist_idx']):
        assert rel_err(hist[idx], z['hist'][k]) < 2e-3, (name, int(idx))


SD_CASES = ['ragged3', 'full2', 'single', 'shuffled']


@pytest.mark.parametrize('tag', SD_CASES)
def test_struct_diffusion_single_evaluation(tag):
    """the transformer baseline: ragged token counts, the all-ones mask of an unpadded graph,
    the head/graph mask mix-up and batch.shuffled are all in these cases"""
    z = golden('struct_diffusion')
    m = oracle_model('qualitative', 64, 'weights_qualitative_h64_sd.npz', model='StructDiffusion')
    g = m.graph(golden_batch(z, tag + '/'))
    for i, t in enumerate(z[tag + '/t']):
        out = g.denoise(z[tag + '/poses'][i], int(t))
        assert rel_err(out, z[tag + '/out'][i]) < 2e-5, (tag, int(t))


def test_struct_diffusion_rejects_long_sequences():
    m = oracle_model('qualitative', 64, 'weights_qualitative_h64_sd.npz', model='StructDiffusion')
    b = worlds.qualitative_batch(1, 8, seed=3)                 # 9 tokens > max_seq_len 8: the reference raises too
    g = m.graph(b)
    out = g.denoise(np.zeros((9, 4), dtype=np.float32), 5)
    assert np.isnan(out).all()


def test_struct_diffusion_chains():
    z = golden('chain_sd64_ula')
    final, hist = _run_oracle_chain(z)
    assert np.abs(final - z['final']).max() < 1e-4
    for k, idx in enumerate(z['hist_idx']):
        assert rel_err(hist[idx], z['hist'][k]) < 2e-3, int(idx)
    # plain ancestral sampling with B = 2 drifts linearly to |x| ~ 1e3 in the reference itself (graph 1 attends
    # with graph 0's pad mask, denoise_fn.py:434): parity is relative there
    z = golden('chain_sd64_noebm')
    final, hist = _run_oracle_chain(z)
    assert np.abs(z['final']).max() > 100.0
    assert rel_err(final, z['final']) < 1e-4
    for k, idx in enumerate(z['hist_idx']):
        assert rel_err(hist[idx], z['hist'][k]) < 1e-4, int(idx)


def test_stability_mode_vs_reference():
    """'stability_flat' (three constraint types): single evaluations and a ULA chain of the reference on a synthetic graph"""
    z = golden('stability')
    W = {k[2:]: z[k] for k in z.files if k.startswith('w/')}
    m = oracle.OracleModel(W, worlds.MODE_DIMS['stability_flat'], 64, 3, timesteps=50, samples_per_step=3)
    g = m.graph(golden_batch(z))
    for i, t in enumerate(z['t']):
        assert rel_err(g.denoise(z['poses'][i], int(t)), z['out'][i]) < 2e-5
    final, hist = g.chain('ULA', seed=int(z['seed']), history=True)
    assert int(z['n_randn']) == 1 + 50 * 4
    for k in range(51):                                   # untrained weights: the chain grows to ~5e3, parity is relative
        assert rel_err(hist[k], z['hist'][k]) < 1e-4, k


def test_robot_energy_mode_vs_reference():
    """energy mode with a grasp group (K_in = 6H): the reference's autograd gradient and batch energy"""
    z = golden('robot_energy')
    W = {k[2:]: z[k] for k in z.files if k.startswith('w/')}
    g = oracle.OracleModel(W, worlds.MODE_DIMS['robot_box'], 64, 2, energy_wrapper=True).graph(golden_batch(z))
    for i, t in enumerate(z['t']):
        grad, E = g.energy_grad(z['poses'][i], int(t))
        assert rel_err(grad, z['grad'][i]) < 5e-5 and abs(E - z['energy'][i]) < 1e-4 * (1 + abs(z['energy'][i]))


def segment_errors(run_segment, z, segments):
    """chains too long for the CPU suite (or that pass a huge transient): the reference's recorded states as starting points.  (a, b) = history
    indices, i.e. timesteps T - 1 - a .. T - b; -> [(a, b, relative error of the state at b)]"""
    T, idx = int(z['T']), list(z['hist_idx'])
    return [(a, b, rel_err(run_segment(z['hist'][idx.index(a)], T - 1 - a, T - b), z['hist'][idx.index(b)])) for a, b in segments]


EPS2_SEGMENTS = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 10), (50, 100)]


def eps2_segment_errors(run_segment, z):
    """ULA on an energy-wrapped model with ebm_per_steps = 2 (Langevin steps on even timesteps only, ddpm.py:330): the
    fixture passes a 1e22 transient, so segments from recorded reference states (odd and even timesteps both covered)"""
    idx = list(z['hist_idx'])
    return [(a, b, rel_err(run_segment(z['hist'][idx.index(a)], 99 - a, 100 - b), z['hist'][idx.index(b)])) for a, b in EPS2_SEGMENTS]


def test_ebm_per_steps_vs_reference():
    z = golden('chain_t64_ula_energy_eps2')
    assert int(z['n_randn']) == 1 + 100 + 50 * 3
    m = oracle.OracleModel(weights('weights_diffuse_pairwise_h64_energy.npz'), worlds.MODE_DIMS['diffuse_pairwise'], 64, 2, timesteps=100,
                           samples_per_step=3, energy_wrapper=True, ebm_per_steps=2)
    g = m.graph(golden_batch(z))
    errs = eps2_segment_errors(lambda x, tf, tl: g.chain('ULA', seed=int(z['seed']), x=x, t_first=tf, t_last=tl), z)
    assert all(e[2] < 1e-4 for e in errs), errs


def test_constructor_options_vs_reference():
    """normalize=False, custom betas, a per-timestep samples_per_step schedule and step_sizes='0.5*self.betas'"""
    z = golden('options')
    m = oracle.OracleModel(weights('weights_qualitative_h64.npz'), worlds.MODE_DIMS['qualitative'], 64, 13, timesteps=60, normalize=False)
    m.set_schedule(betas=z['betas'], step_sizes=z['step_sizes'], samples_per_step=z['sps'])
    s = m.schedule()
    assert np.array_equal(s['step_sizes'], z['step_sizes']) and np.array_equal(s['posterior_log_variance_clipped'], z['posterior_log_variance_clipped'])
    g = m.graph(golden_batch(z))
    for i, t in enumerate(z['t']):
        assert rel_err(g.denoise(z['poses'][i], int(t)), z['out'][i]) < 2e-5
    final, hist = g.chain('ULA', seed=int(z['seed']), history=True)
    assert np.abs(final - z['final']).max() < 1e-4
    for k in range(61):
        assert rel_err(hist[k], z['hist'][k]) < 1e-4, k


MALA_SEGMENTS = [(0, 1), (1, 2), (50, 100), (100, 200), (200, 300), (900, 950), (950, 990), (990, 998),
                 (998, 999), (999, 1000), (900, 1000)]


def mala_segment_errors(run_segment, z, segments=None):
    """MALA (ddpm.py:999-1047) accepts per node with a *discrete* test u < exp(.), and this fixture's
    chain passes through a 1e8 transient in its first ~50 timesteps, so a full chain from the initial
    draw is chaotic (two fp32 runs of the reference itself would not agree).  Parity is therefore
    checked segment by segment from recorded reference states: every timestep's arithmetic (gradient,
    batch-scalar energies, proposal log-probabilities, accept mask) must reproduce the reference's next
    recorded state.  A near-tie may still flip one node's decision: at most one of the
    segments may contain flipped rows, all others must match within 1e-4 (early segments: relative)."""
    idx = list(z['hist_idx'])
    bad = []
    for i0, i1 in (segments or MALA_SEGMENTS):
        k0, k1 = idx.index(i0), idx.index(i1)
        x = run_segment(z['hist'][k0], 999 - i0, 1000 - i1)
        want = z['hist'][k1]
        scale = 1.0 + (np.abs(want).max() if i1 < 50 else 0.0)
        err = np.abs(x - want).max(axis=1)
        rows_off = int((err > 1e-4 * scale).sum())
        if rows_off:
            bad.append((i0, i1, rows_off, float(err.max() / scale)))
    return bad


def test_mala_segments_vs_reference():
    z = golden('chain_t64_mala')
    m = oracle_model('diffuse_pairwise', 64, 'weights_diffuse_pairwise_h64_energy.npz', T=1000, S=int(z['S']), energy=True)
    g = m.graph(golden_batch(z))
    assert int(z['n_rand']) == 1000 * int(z['S'])            # one rand(N) per MALA inner step
    bad = mala_segment_errors(lambda x, tf, tl: g.chain('MALA', seed=int(z['seed']), x=x, t_first=tf, t_last=tl), z)
    assert len(bad) <= 1 and all(b[2] <= 2 for b in bad), bad
    # acceptance bookkeeping (MetropolisSampler, ddpm.py:969-996): rates in [0,1], high at low noise
    _, acc = g.chain('MALA', seed=int(z['seed']), x=z['hist'][list(z['hist_idx']).index(990)], t_first=9, t_last=0, accept=True)
    assert (acc[:10] >= 0).all() and (acc[:10] <= 1).all() and acc[:10].mean() > 0.5


def mala_timestep_errors(run_step, z, timesteps):
    """BASELINE config C4's sampler at its hidden width (chain_t256_mala: 12-triangle graphs, H = 256, MALA S = 10, every
    state of the reference chain recorded together with the reference's own acceptance log, ddpm.py:979-996).  Each
    listed timestep is run from the reference's recorded state; it must reproduce the next recorded state within 1e-4
    (relative to the state's magnitude) AND the reference's mean acceptance of that timestep exactly (a mean over
    N x S binary decisions: any flipped decision shows).  Returns the timesteps where either fails, as
    (t, rows off, acceptance here, acceptance of the reference).  The chain as a whole is not comparable: one flipped
    near-tie u ~ exp(.) changes a node's state by O(1) and every later draw with it (the reference's own fp32 and
    fp64 runs part ways the same way), which is why the fixture records every state.  Inside a timestep a flip also
    moves the batch-scalar energy that all nodes share in the S - 1 inner steps after it, so a flagged timestep can show
    several rows off; what is bounded is the NUMBER of flagged timesteps (callers assert <= 1 %)."""
    T = int(z['T'])
    n_steps = int(z['S_accept']) if 'S_accept' in getattr(z, 'files', z) else int(z['S'])        # (HMC: the reference's fixed 4 inner steps, not samples_per_step)
    tol_acc = 0.25 / (z['x'].shape[0] * n_steps)
    bad = []
    for t in timesteps:
        k = T - 1 - t
        res = run_step(z['hist'][k], t)
        x, acc = res[0], res[1]
        want = z['hist'][k + 1]
        both = np.isfinite(want).all(axis=1) & np.isfinite(x).all(axis=1)
        same_nonfinite = np.array_equal(np.isfinite(want), np.isfinite(x))
        tol = 1e-4
        if 'next_f64' in getattr(z, 'files', z):
            # fixtures that carry the REFERENCE's own fp32-vs-fp64 disagreement over this very timestep (oracle/gen_golden.py ref_single_timestep:
            # the fp64 reference run for one timestep from the same recorded state): where the map of a timestep amplifies rounding differences
            # (HMC's leapfrog on the composed energy at hidden_dim 256: up to 1.2e-2) the bar is 8 x that disagreement, never below 1e-4
            floor = float(np.abs(z['next_f64'][k] - want).max() / (1.0 + np.abs(want).max()))
            tol = max(tol, 8.0 * floor)
        off = int((np.abs(x - want)[both].max(axis=1) > tol * (1.0 + np.abs(want[both]).max())).sum()) if both.any() else 0
        if not same_nonfinite:
            off += int((np.isfinite(want) != np.isfinite(x)).any(axis=1).sum())
        if off or abs(float(acc[t]) - float(z['accept'][t])) > tol_acc:
            rec = (int(t), off, float(acc[t]), float(z['accept'][t]))
            if len(res) > 2 and res[2] is not None:
                # the accept kernels' margins of this timestep (log acceptance ratio - log u per node row and inner step, and the sum of
                # the absolute values of the ratio's terms): a difference from the reference must have BEGUN at a near-tie, i.e. some decision
                # of the timestep has a margin that is within fp32 rounding of the terms it is the difference of.  -> smallest |margin| / scale
                marg = np.asarray(res[2], dtype=np.float64)              # [inner steps, 2, N]: margin, scale of its terms
                rel = np.abs(marg[:, 0]) / np.maximum(marg[:, 1], 1e-30)
                rec = rec + (float(np.nanmin(rel)) if np.isfinite(rel).any() else float('nan'),)
            bad.append(rec)
    return bad


def test_mala_h256_timesteps_vs_reference():
    z = golden('chain_t256_mala')
    assert int(z['H']) == 256 and int(z['S']) == 10 and len(z['hist_idx']) == 1001
    assert int(z['n_rand']) == 1000 * 10 and 0.3 < float(z['accept'].mean()) < 0.9       # a chain that does accept and reject
    m = oracle_model('diffuse_pairwise', 256, 'weights_diffuse_pairwise_h256_energy.npz', T=1000, S=10, energy=True)
    g = m.graph(golden_batch(z))
    ts = list(range(999, -1, -37))                        # 28 of the 1000 timesteps (the GPU test runs them all)
    bad = mala_timestep_errors(lambda x, t: g.chain('MALA', seed=int(z['seed']), x=x, t_first=t, t_last=t, accept=True), z, ts)
    assert len(bad) <= 1 and all(abs(b[2] - b[3]) < 0.05 for b in bad), bad      # (a flipped near-tie may cascade within its timestep)


def test_robot_h256_chain_vs_reference():
    """BASELINE config C5's mode at its hidden width: 10-object robot_box graphs, H = 256, ULA S = 10"""
    z = golden('chain_r256_ula')
    m = oracle_model('robot_box', 256, 'weights_robot_box_h256.npz', T=1000, S=10)
    g = m.graph(golden_batch(z))
    x, hist = g.chain('ULA', seed=int(z['seed']), history=True)
    assert np.abs(x - z['final']).max() < 1e-4
    for k, idx in enumerate(z['hist_idx']):
        assert rel_err(hist[idx], z['hist'][k]) < 1e-4, int(idx)


def test_single_evaluation_h256_energy_and_robot():
    z = golden('single_eval_h256')
    m = oracle_model('diffuse_pairwise', 256, 'weights_diffuse_pairwise_h256_energy.npz', energy=True)
    g = m.graph(golden_batch(z, 't256e/'))
    for i, t in enumerate(z['t256e/t']):
        grad, E = g.energy_grad(z['t256e/poses'][i], int(t))
        assert abs(E - z['t256e/energy'][i]) <= 2e-5 * (1 + abs(z['t256e/energy'][i]))
        assert rel_err(grad, z['t256e/grad'][i]) < 5e-5, t
    m = oracle_model('robot_box', 256, 'weights_robot_box_h256.npz')
    g = m.graph(golden_batch(z, 'r256/'))
    for i, t in enumerate(z['r256/t']):
        assert rel_err(g.denoise(z['r256/poses'][i], int(t)), z['r256/out'][i]) < 2e-5, t


def hmc_T20_errors(run_step, z):
    """AnnealedMUHASampler (ddpm.py:1050-1128) at T = 20, where its proposals do get accepted: every timestep is
    run from the reference's recorded state; returns the timesteps whose next state or mean acceptance differ"""
    T = int(z['T'])
    bad = []
    for k in range(T):
        t = T - 1 - k
        x, acc = run_step(z['hist'][k], t)
        if rel_err(x, z['hist'][k + 1]) > 2e-4 or abs(float(acc[t]) - float(z['accept'][t])) > 1e-6:
            bad.append((t, rel_err(x, z['hist'][k + 1]), float(acc[t]), float(z['accept'][t])))
    return bad


@pytest.mark.parametrize('name', ['chain_t256_hmc_T20', 'chain_t256_hmc_T100'])
def test_hmc_h256_vs_reference(name):
    """the oracle's HMC at hidden_dim 256 against the reference's recorded chains (round 5): every timestep of the T = 20 chain, every fifth of
    the T = 100 one"""
    z = golden(name)
    T = int(z['T'])
    m = oracle_model('diffuse_pairwise', 256, 'weights_diffuse_pairwise_h256_energy.npz', T=T, energy=True)
    g = m.graph(golden_batch(z))
    assert int(z['n_randn']) == 1 + T * 6 and int(z['n_rand']) == T * 4 and np.isfinite(z['hist']).all()
    step = lambda x, t: g.chain('HMC', seed=int(z['seed']), x=x, t_first=t, t_last=t, accept=True)
    if T == 20:
        assert len(set(np.round(z['accept'], 4))) >= 2 and 0.2 < float(z['accept'].mean()) < 0.8
        bad = hmc_T20_errors(step, z)
    else:
        bad = []
        for k in range(0, T, 5):
            t = T - 1 - k
            x, acc = step(z['hist'][k], t)
            if rel_err(x, z['hist'][k + 1]) > 2e-4 or abs(float(acc[t]) - float(z['accept'][t])) > 1e-6:
                bad.append((t, rel_err(x, z['hist'][k + 1]), float(acc[t]), float(z['accept'][t])))
    assert not bad, bad


def test_hmc_vs_reference():
    """SURVEY 8a row 13.  The inner-index-as-timestep quirk makes the leapfrog use step size 2 beta_i and mass
    (9 beta_i)^2 of i in 0..3: at T = 1000 that is a ~600x blow-up of every proposal and the reference never
    accepts one (recorded acceptance is 0 at all 1000 timesteps); at T = 20 acceptance is mixed."""
    z = golden('chain_t64_hmc_T20')
    m = oracle_model('diffuse_pairwise', 64, 'weights_diffuse_pairwise_h64_energy.npz', T=20, energy=True)
    g = m.graph(golden_batch(z))
    assert int(z['n_randn']) == 1 + 20 * 6 and int(z['n_rand']) == 20 * 4
    assert 0 < (z['accept'] > 0).sum() and len(set(np.round(z['accept'], 4))) >= 4     # the fixture exercises partial accepts
    bad = hmc_T20_errors(lambda x, t: g.chain('HMC', seed=int(z['seed']), x=x, t_first=t, t_last=t, accept=True), z)
    assert not bad, bad
    z = golden('chain_t64_hmc')
    assert not z['accept'].any()
    m = oracle_model('diffuse_pairwise', 64, 'weights_diffuse_pairwise_h64_energy.npz', T=1000, energy=True)
    g = m.graph(golden_batch(z))
    acc_seen = []

    def seg(x, tf, tl):
        out, acc = g.chain('HMC', seed=int(z['seed']), x=x, t_first=tf, t_last=tl, accept=True)
        acc_seen.append(acc[tl:tf + 1])
        return out
    bad = mala_segment_errors(seg, z, segments=[(0, 1), (1, 2), (100, 200), (900, 950), (990, 998), (998, 999), (999, 1000)])
    assert not bad, bad
    assert not np.concatenate(acc_seen).any()


def test_fp32_noise_floor_of_the_reference():
    """the reference's own fp32-vs-fp64 disagreement bounds how tight parity can be"""
    z32, z64 = golden('chain_q64_T1000_B4'), golden('chain_q64_T1000_B4_f64')
    floor = np.abs(z32['final'].astype(np.float64) - z64['final']).max()
    assert floor < 1e-5
    o64 = _run_oracle_chain(z32, f64=True, history=False)
    assert np.abs(o64 - z64['final']).max() < 1e-5


def test_torch_proxy_matches_reference():
    """the cost-faithful PyTorch proxy (cpu_baseline 'port') reproduces reference outputs"""
    import torch_proxy
    z = golden('single_eval')
    for tag, mode, H, wfile in [CASES[1], CASES[4]]:
        model = torch_proxy.ProxyDiffuser(weights(wfile), worlds.MODE_DIMS[mode], H, MODE_TYPES[mode])
        b = golden_batch(z, tag + '/')
        for i, t in enumerate(z[tag + '/t'][:3]):
            with torch.no_grad():
                got = model(torch.from_numpy(z[tag + '/poses'][i]), b, torch.tensor([int(t)])).numpy()
            assert rel_err(got, z[tag + '/out'][i]) < 1e-5
    # energy mode (the C4 leg of bench.py's cpu_baseline): autograd gradient and energy vs the reference's
    model = torch_proxy.ProxyDiffuser(weights('weights_diffuse_pairwise_h64_energy.npz'), worlds.MODE_DIMS['diffuse_pairwise'], 64, 2)
    b = golden_batch(z, 't64e/')
    for i, t in enumerate(z['t64e/t'][:3]):
        grad, E = model.energy_grad(torch.from_numpy(z['t64e/poses'][i]), b, torch.tensor([int(t)]))
        assert rel_err(grad.numpy(), z['t64e/grad'][i]) < 1e-5 and abs(float(E) - z['t64e/energy'][i]) <= 1e-5 * (1 + abs(z['t64e/energy'][i]))
    # and one MALA timestep runs (cost model of the baseline: 1 + 3 S forward+backward passes)
    sch = torch_proxy.cosine_schedule(1000)
    g = torch.Generator().manual_seed(0)
    N = b.x.shape[0]
    x = torch_proxy.mala_timestep(model, sch, b, torch.zeros(N, 4), 300, 2, lambda: torch.randn((N, 4), generator=g), lambda: torch.rand((N,), generator=g))
    assert torch.isfinite(x).all()


def test_timestep_restart_equals_full_chain():
    """chain_run(init=0, t_first..t_last) continues a chain exactly (single-timestep vectors)"""
    z = golden('chain_q64_T100_B1')
    m = oracle_model('qualitative', 64, 'weights_qualitative_h64.npz', T=100)
    g = m.graph(golden_batch(z))
    full, hist = g.chain('ULA', seed=int(z['seed']), history=True)
    x = g.chain('ULA', seed=int(z['seed']), x=hist[40], t_first=59, t_last=20)
    assert np.array_equal(x, hist[80])


def test_mala_h128_timesteps_vs_reference():
    z = golden('chain_t128_mala')
    m = oracle_model('diffuse_pairwise', 128, 'weights_diffuse_pairwise_h128_energy.npz', T=int(z['T']), S=int(z['S']), energy=True)
    g = m.graph(golden_batch(z))
    seed = int(z['seed'])
    bad = mala_timestep_errors(lambda x, t: g.chain('MALA', seed=seed, x=x, t_first=t, t_last=t, accept=True), z, range(int(z['T']) - 1, -1, -1))
    assert len(bad) <= 1, bad


def test_reference_recipe_weights_overflow_in_the_reference_sampler():
    """the fixture behind DESIGN.md section 7: the reference's own sampler on weights trained with the reference's recipe as
    written overflows fp32 on every one of sixteen 8-object graphs; the oracle reproduces the recorded transient"""
    from bench import load_weights
    import os
    from conftest import ROOT
    z = golden('chain_q256_ref300k_B16')
    bad = ~np.isfinite(z['final']).all(axis=1)
    assert int(bad.sum()) == 128 and np.abs(z['hist'][1]).max() > 1e8 and np.abs(z['hist'][5]).max() > 1e34
    W = load_weights(os.path.join(ROOT, 'weights', 'qualitative_h256_ref300k.npz'))
    m = oracle.OracleModel({k: v.numpy() if hasattr(v, 'numpy') else v for k, v in W.items()}, worlds.MODE_DIMS['qualitative'], 256, 13,
                           timesteps=1000, samples_per_step=10)
    g = m.graph(golden_batch(z))
    for k in range(3):                                          # timesteps 999, 998, 997 from the recorded states
        x1 = g.chain('ULA', seed=int(z['seed']), x=z['hist'][k], t_first=999 - k, t_last=999 - k)
        assert rel_err(x1, z['hist'][k + 1]) < 2e-3, k


# ---------------------------------------------------------------- round 6: the benchmark width, pinned to the reference

SD256_W = 'weights_qualitative_h256_sd.npz'
SD256_CASES = ['ragged3', 'full2', 'single', 'shuffled', 'ragged8']


@pytest.mark.parametrize('tag', SD256_CASES)
def test_struct_diffusion_h256_single_evaluation(tag):
    """the transformer baseline at the width bench.py --config sd runs (hidden_dim 256 -> transformer width 512), weights trained with the
    reference's loss: single evaluations by the imported reference (transformer.py:43-82, denoise_fn.py:391-451)"""
    z = golden('struct_diffusion_h256')
    g = oracle_model('qualitative', 256, SD256_W, model='StructDiffusion').graph(golden_batch(z, tag + '/'))
    for i, t in enumerate(z[tag + '/t']):
        assert rel_err(g.denoise(z[tag + '/poses'][i], int(t)), z[tag + '/out'][i]) < 2e-5, (tag, int(t))


SD256_SEGMENTS = [(0, 1), (1, 2), (998, 999), (999, 1000)]


def test_struct_diffusion_h256_chain_segments():
    """chain_sd256_ula (T = 1000, ULA S = 10, four ragged graphs): 11 000 evaluations at width 512 are minutes of oracle time, so the CPU suite
    runs single timesteps from the reference's recorded states -- the first two and the last two (the GPU suite runs the whole chain)"""
    z = golden('chain_sd256_ula')
    assert int(z['H']) == 256 and int(z['S']) == 10 and int(z['n_randn']) == 1 + 1000 * 11 and np.abs(z['final']).max() < 5.0
    g = oracle_model('qualitative', 256, SD256_W, T=1000, S=10, model='StructDiffusion').graph(golden_batch(z))
    errs = segment_errors(lambda x, tf, tl: g.chain('ULA', seed=int(z['seed']), x=x, t_first=tf, t_last=tl), z, SD256_SEGMENTS)
    # the bars of the full chains (test_full_chain_final_poses): history checkpoints 2e-3 relative -- the first timesteps' ten Langevin steps have gain
    # > 1 and amplify the fp32 rounding differences of two implementations (measured here: 6e-5, 2.4e-4) --, the states the final poses hang on 1e-4
    assert all(e[2] < (2e-3 if e[1] < 900 else 1e-4) for e in errs), errs


ULAPLUS256_SEGMENTS = [(0, 1), (998, 999), (999, 1000)]


def test_ulaplus_h256_segments():
    """ULA+ (16 / 12 / 8 / 4 Langevin steps by quarter of the schedule, ddpm.py:297-299) at hidden_dim 256: the first timestep (16 steps) and the
    last two (4 steps) from the reference's recorded states; the draw count pins the step counts of all four quarters"""
    z = golden('chain_q256_ulaplus')
    assert int(z['n_randn']) == 1 + 1000 + 250 * (16 + 12 + 8 + 4) and np.abs(z['hist']).max() > 1e15
    m = oracle_model('qualitative', 256, 'weights_qualitative_h256.npz', T=1000)
    g = m.graph(golden_batch(z))
    errs = segment_errors(lambda x, tf, tl: g.chain('ULA+', seed=int(z['seed']), x=x, t_first=tf, t_last=tl), z, ULAPLUS256_SEGMENTS)
    assert all(e[2] < 1e-4 for e in errs), errs


def chaotic_timestep_errors(run_step, z, timesteps):
    """A chain that is chaotic IN THE REFERENCE (chain_t256_ula_energy_eps2: its own fp32 and fp64 runs are unrelated half-way) is compared timestep by
    timestep from the reference's recorded states; the fixture carries, per timestep, the reference's own fp64 successor of the recorded fp32 state
    (next_f64, oracle/gen_golden.py ref_single_timestep), and a timestep's bar is max(1e-4, 8 x the reference's own fp32-vs-fp64 disagreement over it)
    -- the rule of the HMC fixtures.  -> [(t, error, bar)] of the timesteps over their bar"""
    T = int(z['T'])
    bad = []
    for t in timesteps:
        k = T - 1 - t
        want = z['hist'][k + 1]
        bar = max(1e-4, 8.0 * rel_err(z['next_f64'][k], want))
        err = rel_err(run_step(z['hist'][k], t), want)
        if not err < bar:
            bad.append((int(t), err, bar))
    return bad


def assert_only_transient_timesteps_flagged(z, bad, limit):
    """chain_t256_ula_energy_eps2 between timesteps 190 and 160 is the decay of a 2e22 transient: every Langevin step leaves a remainder orders of
    magnitude smaller than the state it started from (x - 2 kappa eps(x) with eps ~ x / 2 kappa: cancellation), which amplifies the ~1e-7 by which two
    correct fp32 gradient evaluations differ by the same orders -- per implementation, in a few rows of a few timesteps (measured: the fp32 oracle at
    t = 176 and 184, 8e-2 and 2e-4 on two rows each, where its single gradient evaluations agree with its fp64 build to 8e-7 and the fp64 oracle meets
    the reference to 9e-7).  The reference's own fp32-vs-fp64 pair is ONE sample of that amplification, so a handful of timesteps may exceed 8 x it:
    they must all lie inside the transient (|x| > 1e6), where nothing downstream depends on them at the 1e-4 level (the chain forgets: final poses
    of the fp32 and fp64 reference runs differ by 4e-3 anyway)."""
    T = int(z['T'])
    assert len(bad) <= limit and all(np.abs(z['hist'][T - 1 - t]).max() > 1e6 for t, _, _ in bad), bad


def eps2_h256_model(z, sampler_steps):
    return oracle.OracleModel(weights('weights_diffuse_pairwise_h256_energy.npz'), worlds.MODE_DIMS['diffuse_pairwise'], 256, 2, timesteps=int(z['T']),
                              samples_per_step=sampler_steps, energy_wrapper=True, ebm_per_steps=2)


def test_ebm_per_steps_h256_vs_reference():
    """ebm_per_steps = 2 (ddpm.py:330) at hidden_dim 256, energy-mode ULA: Langevin steps on even timesteps only"""
    z = golden('chain_t256_ula_energy_eps2')
    assert int(z['T']) == 200 and int(z['n_randn']) == 1 + 200 + 100 * 5 and len(z['hist_idx']) == 201
    g = eps2_h256_model(z, 5).graph(golden_batch(z))
    ts = list(range(199, 179, -1)) + list(range(179, -1, -7)) + [1, 0]        # the transient's first twenty timesteps, then every seventh (odd and even)
    bad = chaotic_timestep_errors(lambda x, t: g.chain('ULA', seed=int(z['seed']), x=x, t_first=t, t_last=t), z, ts)
    assert_only_transient_timesteps_flagged(z, bad, 3)


def test_mala_ebm_per_steps_h256_every_timestep_vs_reference():
    """... and under MALA (S = 4): every timestep from the reference's recorded state, the reference's acceptance log reproduced; odd timesteps
    run no inner step (acceptance 0 in the log and here)"""
    z = golden('chain_t256_mala_eps2')
    assert int(z['T']) == 100 and int(z['n_rand']) == 50 * 4 and (z['accept'][1::2] == 0).all() and z['accept'][0::2].max() > 0.5
    g = eps2_h256_model(z, 4).graph(golden_batch(z))
    seed = int(z['seed'])
    bad = mala_timestep_errors(lambda x, t: g.chain('MALA', seed=seed, x=x, t_first=t, t_last=t, accept=True), z, range(99, -1, -1))
    assert len(bad) <= 1, bad
