"""The variants that lost their same-call A/Bs (profiles/r0*_findings.md), compiled only into the EXPERIMENTS build
(-DCCSP_EXPERIMENTS -> csrc/libccsp_hip_exp.so): the one-launch fused evaluation, row GEMM MODEs 1/2/3/5/7, the node update in the edge
kernel's tail (two forms), hipGraph replay, the streamed node kernel, the separate row-sum launch, the bf16x3 alternates.  Each must stay
bitwise equal to / within the bars of the product path.  Outside the default `-m gpu` run:

    CCSP_EXPERIMENTS=1 python -m pytest tests -m gpu_experiments -q        (builds libccsp_hip_exp.so on first use: ~70 s of hipcc)
"""
import numpy as np
import pytest
import torch

from conftest import golden, golden_batch, rel_err, weights, worlds
from diffusion_ccsp_amd import _lib
from test_hip_parity import _golden_chain_model, hip_model

pytestmark = [pytest.mark.gpu_experiments,
              pytest.mark.skipif(not _lib.EXPERIMENTS, reason='needs the experiments build: run with CCSP_EXPERIMENTS=1')]


@pytest.mark.parametrize('env', [{'CCSP_ROW_MODE': '1', 'CCSP_EDGE_MT': '2'}, {'CCSP_ROW_MODE': '2', 'CCSP_EDGE_MT': '1'}, {'CCSP_ROW_MODE': '3', 'CCSP_EDGE_MT': '2'},
                                 {'CCSP_ROW_MODE': '5', 'CCSP_EDGE_MT': '1'}, {'CCSP_ROW_MODE': '7', 'CCSP_EDGE_MT': '1'}, {'CCSP_ROW_MODE': '9'}, {'CCSP_NODE': 'stream'},
                                 {'CCSP_ENERGY_ROWSUM': 'kernel'}])
def test_f16x2_experimental_variants_meet_the_same_bars(device, monkeypatch, env):
    """(see tests/test_hip_parity.py::test_f16x2_residency_variants_meet_the_same_bars) the f16x2 kernels pick a variant by tile count (row GEMM: MODE 0 three workgroups per CU register-staged on 128-row tiles, MODE 6 the
    same staging on 64-row tiles, MODE 4 a ring of LDS stages with counted waits on 64-row tiles for short tile lists; edge kernel: 64-, 32- or
    16-edge tiles; pose encoder on the fp32 pipe, energy backward on the bf16x3 kernels; the generic node kernel); each is forced here on the
    same inputs: single evaluations, a full chain and energy-mode gradients at H = 256.  (The variants that lost their A/Bs -- MODEs 1/2/3/5/7,
    the streamed node kernel, the separate row-sum launch -- live in the experiments build: tests/test_experiments.py.)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    z = golden('single_eval')
    b = golden_batch(z, 'q256/')
    den, _ = hip_model(device, 'qualitative', 256, 'weights_qualitative_h256.npz')
    for i, t in enumerate(z['q256/t']):
        got = den(torch.from_numpy(z['q256/poses'][i]), b, torch.tensor([int(t)]), eval=True).cpu().numpy()
        assert rel_err(got, z['q256/out'][i]) < 2e-5, (env, int(t))
    zc = golden('chain_q256_T1000_B4')
    den, gd, bc = _golden_chain_model(device, zc)
    assert np.abs(gd.sample(bc, seed=int(zc['seed'])).cpu().numpy() - zc['final']).max() < 1e-4
    z = golden('single_eval_h256')
    den, gd = hip_model(device, 'diffuse_pairwise', 256, 'weights_diffuse_pairwise_h256_energy.npz', EBM='MALA', energy=True)
    b = golden_batch(z, 't256e/')
    for i, t in enumerate(z['t256e/t']):
        grad, E = den(torch.from_numpy(z['t256e/poses'][i]), b, torch.tensor([int(t)]), tag='EBM')
        assert rel_err(grad.cpu().numpy(), z['t256e/grad'][i]) < 5e-5, (env, int(t))
        assert abs(float(E) - z['t256e/energy'][i]) <= 2e-5 * (1 + abs(z['t256e/energy'][i]))



@pytest.mark.parametrize('env', [{'CCSP_ROW_TILE': '64'}, {'CCSP_EDGE_KERNEL': '1'}, {'CCSP_ROW_TILE': '64', 'CCSP_EDGE_KERNEL': '1', 'CCSP_LANES': '1'}])
def test_alternate_kernels_meet_the_same_bars(device, monkeypatch, env):
    """the kernels behind the runtime switches of INTEGRATION.md section 4 (64-row row GEMM, single-stage edge kernel) stay
    correct: single evaluations, a full chain and the energy gradient at hidden_dim 256"""
    monkeypatch.setenv('CCSP_MMA', 'bf16x3')                # (these switches select among the bf16x3 kernels)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    z = golden('single_eval')
    b = golden_batch(z, 'q256/')
    den, _ = hip_model(device, 'qualitative', 256, 'weights_qualitative_h256.npz')
    for i, t in enumerate(z['q256/t']):
        got = den(torch.from_numpy(z['q256/poses'][i]), b, torch.tensor([int(t)]), eval=True).cpu().numpy()
        assert rel_err(got, z['q256/out'][i]) < 2e-5, (env, int(t))
    zc = golden('chain_q256_T100_B1')
    den, gd, bc = _golden_chain_model(device, zc)
    assert np.abs(gd.sample(bc, seed=int(zc['seed'])).cpu().numpy() - zc['final']).max() < 1e-4
    import oracle
    from diffusion_ccsp_amd import ConstraintDiffuser
    e = ConstraintDiffuser(dims=worlds.MODE_DIMS['diffuse_pairwise'], hidden_dim=256, input_mode='diffuse_pairwise', EBM='MALA',
                           energy_wrapper=True, device=device, verbose=False)
    e.reset_parameters(2)
    W = {k: v.cpu().numpy() for k, v in e.state_dict().items()}
    tb = worlds.triangular_batch(3, 6, seed=4).to_torch()
    og = oracle.OracleModel(W, worlds.MODE_DIMS['diffuse_pairwise'], 256, 2, energy_wrapper=True).graph(tb)
    poses = (np.random.default_rng(0).standard_normal((tb.x.shape[0], 4)) * 0.5).astype(np.float32)
    grad, en = e(torch.from_numpy(poses), tb, torch.tensor([300]), tag='EBM')
    want, E = og.energy_grad(poses, 300)
    assert rel_err(grad.cpu().numpy(), want) < 5e-5 and abs(float(en) - E) < 1e-4 * (1 + abs(E))



def test_hipgraph_mode_is_bitwise_identical(device, monkeypatch):
    """CCSP_GRAPH=1: small batches replay one captured hipGraph per timestep, the step-dependent scalars come from a
    device table (StepEntry) -- same kernels, same arithmetic, so results must be bit-equal to plain launches,
    for seeded and injected noise, histories, ULA+ (several graph shapes) and timestep segments"""
    from diffusion_ccsp_amd import noise
    out = {}
    for gm in ('0', '1'):
        monkeypatch.setenv('CCSP_GRAPH', gm)
        res = []
        for name in ('chain_q64_T100_B1', 'chain_q64_ulaplus', 'chain_q256_T100_B1', 'chain_q64_noebm'):
            z = golden(name)
            den, gd, b = _golden_chain_model(device, z)
            x, hist = gd.sample(b, return_history=True, seed=int(z['seed']))
            assert np.abs(x.cpu().numpy() - z['final']).max() < 1e-4, (gm, name)
            res += [x.cpu().numpy(), torch.stack(hist).cpu().numpy()]
            stream = torch.from_numpy(noise.normal_stream(int(z['seed']), gd.n_normal_calls(), b.x.shape[0], gd.dims[-1][0]))
            res.append(gd.sample(b, noise=stream).cpu().numpy())
            T = int(z['T'])
            res.append(gd.p_sample_segment(b, torch.from_numpy(z['hist'][1]), T - 2, T - 5, seed=int(z['seed'])).cpu().numpy())
            res.append(gd.sample(b, seed=5).cpu().numpy())          # second chain on the same graph handle: the captured graphs are reused
        out[gm] = res
    for a, c in zip(out['0'], out['1']):
        assert np.array_equal(a, c, equal_nan=True)



def test_fused_eval_is_bitwise_identical(device, monkeypatch):
    """CCSP_EVAL=fused / fused8 (csrc/ccsp_fused.h: two 256-thread workgroups per CU on 28-row tiles / one persistent 512-thread
    workgroup per CU on 32-row tiles): the row GEMM and the edge decoder as ONE launch per evaluation, the U rows of a
    (type, edge run, output half) tile kept in LDS.  Same operands, exponents, MFMA order and epilogue arithmetic as
    k_rowgemm_h2 + k_edge_h2<false, ., 0> (forced by CCSP_EDGE_MT=2: the small-batch edge kernels sum the second decoder layer
    in another order), so single evaluations and whole chains must be BITWISE those of the two-launch path -- full tiles
    (200 x 8 objects), partial tiles, an irregular graph (isolated node, unknown edge type), shuffled edges (rows shared between
    tiles), the robot model (pose_dim 5) and 12-triangle graphs (66-edge runs) -- and meet the reference goldens."""
    outs = {}
    z = golden('single_eval')
    zr = golden('single_eval_h256')
    for mode_env in ({'CCSP_EVAL': 'split', 'CCSP_EDGE_MT': '2'}, {'CCSP_EVAL': 'fused'}, {'CCSP_EVAL': 'fused8'}):
        for k in ('CCSP_EVAL', 'CCSP_EDGE_MT'):
            monkeypatch.delenv(k, raising=False)
        for k, v in mode_env.items():
            monkeypatch.setenv(k, v)
        res = []
        den1, _ = hip_model(device, 'qualitative', 256, 'weights_qualitative_h256.npz')
        b = golden_batch(z, 'q256/')
        for i, t in enumerate(z['q256/t']):
            got = den1(torch.from_numpy(z['q256/poses'][i]), b, torch.tensor([int(t)]), eval=True).cpu().numpy()
            assert rel_err(got, z['q256/out'][i]) < 2e-5, (mode_env, int(t))
            res.append(got)
        den, gd = hip_model(device, 'qualitative', 256, 'weights_qualitative_h256.npz', T=60, S=4)
        for n_graphs, n_obj in ((200, 8), (7, 5), (33, 3)):
            bb = worlds.qualitative_batch(n_graphs, n_obj, seed=3).to_torch()
            x, hist = gd.sample(bb, seed=11, return_history=True)
            res.append(torch.stack(hist).cpu().numpy())
        q = worlds.qualitative_batch(5, 4, seed=9)
        ei = np.concatenate([q.edge_index, np.array([[3], [4]])], axis=1)
        ea = np.concatenate([q.edge_attr, np.array([99.0], dtype=np.float32)])
        xx = np.concatenate([q.x, q.x[-1:]], axis=0)
        mm = np.concatenate([q.mask, np.zeros(1, dtype=q.mask.dtype)])
        res.append(gd.sample(worlds.GraphBatch(x=xx, edge_index=ei, edge_attr=ea, mask=mm).to_torch(), seed=5).cpu().numpy())
        q = worlds.qualitative_batch(40, 6, seed=4)
        perm = np.random.RandomState(1).permutation(q.edge_index.shape[1])
        res.append(gd.sample(worlds.GraphBatch(x=q.x, edge_index=q.edge_index[:, perm], edge_attr=q.edge_attr[perm], mask=q.mask).to_torch(),
                             seed=6).cpu().numpy())
        denr1, _ = hip_model(device, 'robot_box', 256, 'weights_robot_box_h256.npz')
        denr, gdr = hip_model(device, 'robot_box', 256, 'weights_robot_box_h256.npz', T=40, S=3)
        br = golden_batch(zr, 'r256/')
        for i, t in enumerate(zr['r256/t']):
            got = denr1(torch.from_numpy(zr['r256/poses'][i]), br, torch.tensor([int(t)]), eval=True).cpu().numpy()
            assert rel_err(got, zr['r256/out'][i]) < 2e-5, (mode_env, 'robot', int(t))
            res.append(got)
        res.append(gdr.sample(worlds.robot_box_batch(20, 10, seed=2).to_torch(), seed=3).cpu().numpy())
        dent, gdt = hip_model(device, 'diffuse_pairwise', 256, 'weights_diffuse_pairwise_h256_energy.npz', T=40, S=3)
        res.append(gdt.sample(worlds.triangular_batch(9, 12, seed=2).to_torch(), seed=3).cpu().numpy())
        outs[mode_env['CCSP_EVAL']] = res
    for form in ('fused', 'fused8'):
        for i, (a, c) in enumerate(zip(outs['split'], outs[form])):
            assert a.shape == c.shape and np.array_equal(a, c, equal_nan=True), (form, i, float(np.nanmax(np.abs(a - c))))



def test_row_gemm_direct_fragments_is_bitwise_identical(device, monkeypatch):
    """CCSP_ROW_MODE=7 (k_rowgemm_h2d, csrc/ccsp_fused.h): the row GEMM with the tile's pose-embedding planes resident in LDS and the
    weight fragments streamed straight from global memory in MFMA operand order -- same operands, exponents and MFMA order as the
    staged modes, so U, umax and with them whole chains are bitwise those of MODE 0: full and partial tiles, two lanes, an
    irregular graph, the robot model, energy-mode gradients (the forward row GEMM of the energy path is the same kernel)"""
    outs = {}
    ze = golden('single_eval_h256')
    for mode in ('0', '7'):
        monkeypatch.setenv('CCSP_ROW_MODE', mode)
        res = []
        den, gd = hip_model(device, 'qualitative', 256, 'weights_qualitative_h256.npz', T=60, S=4)
        for n_graphs, n_obj in ((200, 8), (7, 5), (33, 3)):
            x, hist = gd.sample(worlds.qualitative_batch(n_graphs, n_obj, seed=3).to_torch(), seed=11, return_history=True)
            res.append(torch.stack(hist).cpu().numpy())
        q = worlds.qualitative_batch(5, 4, seed=9)
        ei = np.concatenate([q.edge_index, np.array([[3], [4]])], axis=1)
        ea = np.concatenate([q.edge_attr, np.array([99.0], dtype=np.float32)])
        xx = np.concatenate([q.x, q.x[-1:]], axis=0)
        mm = np.concatenate([q.mask, np.zeros(1, dtype=q.mask.dtype)])
        res.append(gd.sample(worlds.GraphBatch(x=xx, edge_index=ei, edge_attr=ea, mask=mm).to_torch(), seed=5).cpu().numpy())
        denr, gdr = hip_model(device, 'robot_box', 256, 'weights_robot_box_h256.npz', T=40, S=3)
        res.append(gdr.sample(worlds.robot_box_batch(20, 10, seed=2).to_torch(), seed=3).cpu().numpy())
        dene, _ = hip_model(device, 'diffuse_pairwise', 256, 'weights_diffuse_pairwise_h256_energy.npz', EBM='MALA', energy=True)
        be = golden_batch(ze, 't256e/')
        for i, t in enumerate(ze['t256e/t']):
            grad, E = dene(torch.from_numpy(ze['t256e/poses'][i]), be, torch.tensor([int(t)]), tag='EBM')
            assert rel_err(grad.cpu().numpy(), ze['t256e/grad'][i]) < 5e-5, (mode, int(t))
            res += [grad.cpu().numpy(), np.asarray(float(E))]
        outs[mode] = res
    for i, (a, c) in enumerate(zip(outs['0'], outs['7'])):
        assert a.shape == c.shape and np.array_equal(a, c, equal_nan=True), i



def test_fused_node_update_is_bitwise_identical(device, monkeypatch):
    """CCSP_FUSE_NODE=1 (opt-in, measured slower: profiles/r03_findings.md): the node update runs in the tail of the edge kernel, by
    the workgroup that delivers a 16-node block's last edge outputs (write-through stores, arrival counters, no barrier).  Same
    arithmetic in the same order: chains must be bitwise those of the three-launch form -- 32-edge tiles (two lanes), 16-edge tiles
    (small batch), isolated nodes (blocks no edge reaches), a history, a chain cut in two."""
    outs = {}
    for flag in ('0', '1', '2'):
        # ('2', round 4: the NODE-GROUPED edge tiles -- a workgroup's rows are the CSR entries of its own run of nodes and their update is its
        # tail, no hand-over between workgroups; k_edge_h2<.., NG>)
        monkeypatch.setenv('CCSP_FUSE_NODE', flag)
        den, gd = hip_model(device, 'qualitative', 256, 'weights_qualitative_h256.npz', T=60, S=4)
        res = []
        for n_graphs, n_obj in ((200, 8), (7, 5)):
            b = worlds.qualitative_batch(n_graphs, n_obj, seed=3).to_torch()
            x, hist = gd.sample(b, seed=11, return_history=True)
            res.append(torch.stack(hist).cpu().numpy())
        z = golden('single_eval')                                    # q64small-style irregular graph: an isolated node, an unknown edge type
        bb = worlds.qualitative_batch(5, 4, seed=9)
        ei = np.concatenate([bb.edge_index, np.array([[3], [4]])], axis=1)
        ea = np.concatenate([bb.edge_attr, np.array([99.0], dtype=np.float32)])
        xx = np.concatenate([bb.x, bb.x[-1:]], axis=0)               # one more node that no edge touches
        mm = np.concatenate([bb.mask, np.zeros(1, dtype=bb.mask.dtype)])
        irr = worlds.GraphBatch(x=xx, edge_index=ei, edge_attr=ea, mask=mm).to_torch()
        res.append(gd.sample(irr, seed=5).cpu().numpy())
        x_mid = gd.p_sample_segment(b, torch.from_numpy(res[1][20]), 39, 20, seed=11).cpu().numpy()
        res.append(x_mid)
        outs[flag] = res
    for flag in ('1', '2'):
        for a, b_ in zip(outs['0'], outs[flag]):
            assert np.array_equal(a, b_, equal_nan=True), flag
        assert np.array_equal(outs[flag][3], outs[flag][1][40], equal_nan=True)



def test_energy_launch_forms_are_bitwise_identical(device, monkeypatch):
    """round 4's launch structure of the energy mode: the update that consumes the gradient inside the gradient evaluation's
    last kernel (against CCSP_ENERGY_NODE=split: two launches) and the decoder backward with pose_dim 4 at compile time (against
    CCSP_ENERGY_BWD_P=generic) change WHERE arithmetic happens, not the arithmetic: MALA chains (reuse on and off), energy-mode ULA
    chains, histories and acceptance rates must be bitwise equal.  (The row sums inside the decoder backward, CCSP_ENERGY_ROWSUM, change
    the association of the sums and are held to the parity bars instead: test_f16x2_residency_variants_meet_the_same_bars.)"""
    outs = []
    for env in ({}, {'CCSP_ENERGY_NODE': 'split'}, {'CCSP_ENERGY_BWD_P': 'generic'}, {'CCSP_ENERGY_NODE': 'split', 'CCSP_MALA_REUSE': '0'}):
        for k in ('CCSP_ENERGY_NODE', 'CCSP_ENERGY_BWD_P', 'CCSP_MALA_REUSE'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        res = []
        for ebm in ('MALA', 'ULA'):
            den, gd = hip_model(device, 'diffuse_pairwise', 256, 'weights_diffuse_pairwise_h256_energy.npz', T=40, S=5, EBM=ebm, energy=True)
            for n_graphs, n_obj, seed in ((24, 12, 5), (1, 6, 9)):
                b = worlds.triangular_batch(n_graphs, n_obj, seed=seed).to_torch()
                x, hist = gd.sample(b, seed=23, return_history=True)
                res += [x.cpu().numpy(), torch.stack(hist).cpu().numpy()]
                if ebm == 'MALA':
                    res.append(gd.last_accept_rates.cpu().numpy())
        outs.append(res)
    assert np.isfinite(outs[0][0]).all()
    for other in outs[1:]:
        assert len(other) == len(outs[0])
        for a, c in zip(outs[0], other):
            assert np.array_equal(a, c, equal_nan=True)




def test_relay_mode_is_bitwise_identical(device, monkeypatch):
    """CCSP_RELAY=1 (round 5, csrc Gate; profiles/r05_findings.md section 5): on small batches the three kernels of an evaluation go to three
    streams and are handed over through device counters (a workgroup polls its producer's counter at entry, adds to its own at exit, with an
    agent-scope release / acquire around it) instead of stream order.  Same kernels, same arithmetic: bit-equal chains, histories and segments --
    one lane and two, a batch above the slot budget (falls back to stream order), and no gate may time out (a fault turns the state into NaN)."""
    out = {}
    for relay in ('0', '1'):
        monkeypatch.setenv('CCSP_RELAY', relay)
        res = []
        for name in ('chain_q256_T100_B1', 'chain_q256_T1000_B4'):
            z = golden(name)
            den, gd, b = _golden_chain_model(device, z)
            x, hist = gd.sample(b, return_history=True, seed=int(z['seed']))
            assert np.abs(x.cpu().numpy() - z['final']).max() < 1e-4, (relay, name)
            res += [x.cpu().numpy(), torch.stack(hist).cpu().numpy()]
            T = int(z['T'])
            res.append(gd.p_sample_segment(b, torch.from_numpy(z['hist'][1]), T - 2, T - 5, seed=int(z['seed'])).cpu().numpy())
        for lanes, graphs in (('1', 24), ('2', 48), ('1', 200)):
            monkeypatch.setenv('CCSP_LANES', lanes)
            monkeypatch.setenv('CCSP_LANE_MIN_EDGES', '1')
            den, gd = hip_model(device, 'qualitative', 256, 'weights_qualitative_h256.npz', T=30, S=3)
            x = gd.sample(worlds.qualitative_batch(graphs, 8, seed=3).to_torch(), seed=11).cpu().numpy()
            assert np.isfinite(x).all()
            res.append(x)
        out[relay] = res
    for a, c in zip(out['0'], out['1']):
        assert np.array_equal(a, c, equal_nan=True)


def test_struct_diffusion_wide_tiles_vs_oracle(device, monkeypatch):
    """round 6, CCSP_SD_TILE=wide (measured slower, experiments build): from 384 token rows on the transformer's in_proj / c_fc / c_proj GEMMs run on
    128 x 128 tiles (k_sd_gemm_h2w; c_proj as four K slices), with the next chunk's staging between the MFMA pairs (CCSP_SD_PIPE) or behind them.
    The reference-generated fixtures have at most 64 token rows, so the wide kernels meet the oracle here -- 64 ragged graphs (512 token rows), the
    reference-trained H = 256 weights, single evaluations at 2e-5 -- next to the narrow kernels on the same inputs (CCSP_SD_TILE=narrow)."""
    sizes = [1 + (3 * k) % 7 for k in range(64)]
    from test_hip_parity import SD256_W, sd_batch
    from conftest import oracle_model
    b = sd_batch(sizes, 77).to_torch()
    og = oracle_model('qualitative', 256, SD256_W, model='StructDiffusion').graph(b)
    rng = np.random.default_rng(5)
    poses = [(rng.standard_normal((b.x.shape[0], 4)) * 0.7).astype(np.float32) for _ in range(2)]
    want = [og.denoise(p, t) for p, t in zip(poses, (3, 700))]
    outs = {}
    for tile, pipe in (('wide', '1'), ('wide', '0'), ('narrow', '1'), ('wide8', '1')):       # (wide8: the same tiles on eight waves, k_sd_gemm_h2x)
        monkeypatch.setenv('CCSP_SD_TILE', tile)
        monkeypatch.setenv('CCSP_SD_PIPE', pipe)          # (the staging of the next chunk between the MFMA pairs, or behind them: same products, same order)
        den, _ = hip_model(device, 'qualitative', 256, SD256_W, model='StructDiffusion')
        outs[tile, pipe] = [den(torch.from_numpy(p), b, torch.tensor([t]), eval=True).cpu().numpy() for p, t in zip(poses, (3, 700))]
        for got, w in zip(outs[tile, pipe], want):
            assert rel_err(got, w) < 2e-5, (tile, pipe)
    assert np.array_equal(np.stack(outs['wide', '1']), np.stack(outs['wide', '0']))
    assert rel_err(np.stack(outs['wide', '1']), np.stack(outs['narrow', '1'])) < 1e-5
    assert np.array_equal(np.stack(outs['wide8', '1']), np.stack(outs['wide', '1']))          # same products in the same order per accumulator
