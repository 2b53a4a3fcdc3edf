"""Host-object contracts of the drop-in classes that are not arithmetic: handle lifetimes (a graph may
outlive its model), the per-batch graph cache, and the argument checks of ccsp_schedule_set."""
import ctypes as C
import gc

import numpy as np
import pytest
import torch

from conftest import weights, worlds

pytestmark = pytest.mark.gpu


def _model(device, T=20, S=2, H=64):
    from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion
    den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=H, input_mode='qualitative', EBM='ULA',
                             device=device, verbose=False)
    den.load_state_dict(weights('weights_qualitative_h%d.npz' % H))
    return den, GaussianDiffusion(den, timesteps=T, EBM='ULA', samples_per_step=S)


def test_graph_may_outlive_its_model(device, monkeypatch):
    """sample a batch that runs as concurrent lanes (child graphs on model-owned streams), reload the weights
    (destroys the native model), sample again, then drop everything in the 'wrong' order"""
    from diffusion_ccsp_amd import _lib
    monkeypatch.setenv('CCSP_LANE_MIN_EDGES', '0')
    den, gd = _model(device)
    b = worlds.qualitative_batch(40, 6, seed=3).to_torch()
    x1 = gd.sample(b, seed=1)
    g_old = gd._last_graph
    assert g_old.h
    den.load_state_dict(weights('weights_qualitative_h64.npz'))      # _drop_handle: graphs, then the model
    assert g_old.h is None                                            # destroyed with the model it was built on
    x2 = gd.sample(b, seed=1)
    assert torch.equal(x1, x2)
    # raw handles: destroy the model first, the graph afterwards (ccsp_model_destroy orphans it)
    L = _lib.lib()
    g = den._graph(b)
    h_graph, h_model = g.h, den._h
    g.h = None
    den._h = None
    den._graphs.clear()
    L.ccsp_model_destroy(h_model)
    L.ccsp_graph_destroy(h_graph)
    torch.cuda.synchronize()
    del gd, den
    gc.collect()


def test_graph_cache_follows_the_batch_object(device):
    den, gd = _model(device)
    outs = []
    for seed in (5, 6, 5):
        b = worlds.qualitative_batch(2, 4, seed=seed).to_torch()      # CPU tensors: the library works on copies
        outs.append(den(torch.zeros(b.x.shape[0], 4), b, torch.tensor([3]), eval=True).cpu().numpy())
        del b
        gc.collect()
    assert np.array_equal(outs[0], outs[2]) and not np.array_equal(outs[0], outs[1])
    assert len(den._graphs) == 0                                      # entries die with their batch
    b = worlds.qualitative_batch(2, 4, seed=5).to_torch()
    g1 = den._graph(b)
    assert den._graph(b) is g1
    b.x[1, 0] += 0.25                                                 # in-place edit of the geometry -> new tables
    g2 = den._graph(b)
    assert g2 is not g1
    want = b.clone()
    assert np.array_equal(den(torch.zeros(b.x.shape[0], 4), b, torch.tensor([3]), eval=True).cpu().numpy(),
                          den(torch.zeros(b.x.shape[0], 4), want, torch.tensor([3]), eval=True).cpu().numpy())
    b.edge_attr = b.edge_attr.clone()                                 # replaced tensor -> new tables
    assert den._graph(b) is not g2


def test_schedule_set_validates_its_arguments(device):
    from diffusion_ccsp_amd import CcspError, GaussianDiffusion, _lib
    den, gd = _model(device, T=20)
    L = _lib.lib()
    h = gd._handle()
    ok = np.full(20, 3, dtype=np.int32)
    assert L.ccsp_schedule_set(h, 20, None, None, ok.ctypes.data, 0) == 0
    assert L.ccsp_schedule_set(h, 19, None, None, ok.ctypes.data, 0) != 0 and b'timesteps' in L.ccsp_last_error()
    bad = ok.copy()
    bad[7] = -1
    assert L.ccsp_schedule_set(h, 20, None, None, bad.ctypes.data, 0) != 0
    bad[7] = 10 ** 7
    assert L.ccsp_schedule_set(h, 20, None, None, bad.ctypes.data, 0) != 0
    betas = np.full(20, 0.1)
    betas[3] = 1.5
    assert L.ccsp_schedule_set(h, 20, betas.ctypes.data, None, None, 2) != 0
    assert L.ccsp_schedule_set(h, 20, None, None, None, -4) != 0
    with pytest.raises(ValueError):
        GaussianDiffusion(den, timesteps=20, EBM='ULA', samples_per_step=torch.ones(19, dtype=torch.int32))
    with pytest.raises(CcspError):
        GaussianDiffusion(den, timesteps=20, EBM='ULA', samples_per_step=-1)
    # two diffusion objects of different lengths on one denoiser: each call re-binds the native model to its own length
    ga = GaussianDiffusion(den, timesteps=20, EBM='ULA', samples_per_step=2)
    gb = GaussianDiffusion(den, timesteps=30, EBM='ULA', samples_per_step=1)
    b = worlds.qualitative_batch(2, 3, seed=1).to_torch()
    xa = ga.sample(b, seed=2)
    assert ga.chain_stats()['evals'] == 20 * 3
    xb = gb.sample(b, seed=2)
    assert gb.chain_stats()['evals'] == 30 * 2
    assert torch.equal(ga.sample(b, seed=2), xa) and torch.equal(gb.sample(b, seed=2), xb)


def test_mala_energy_hook_is_called_on_the_chain_stream(device):
    """MALA global-batch mode (ccsp_model_set_energy_hook through sharding.enable_global_batch_energy): with a one-rank
    'reduction' the chain is bit-equal to the hook-free run; a reduction that changes the energies changes the accept
    decisions; removing the hook restores the first result"""
    from diffusion_ccsp_amd import ComposedEBMDenoiseFn, ConstraintDiffuser, GaussianDiffusion, sharding
    den = ConstraintDiffuser(dims=worlds.MODE_DIMS['diffuse_pairwise'], hidden_dim=64, input_mode='diffuse_pairwise', EBM='MALA',
                             energy_wrapper=True, device=device, verbose=False)
    den.load_state_dict(weights('weights_diffuse_pairwise_h64_energy.npz'))
    gd = GaussianDiffusion(ComposedEBMDenoiseFn(den), timesteps=1000, EBM='MALA', samples_per_step=3)
    b = worlds.triangular_batch(3, 6, seed=19).to_torch()
    x0 = torch.zeros(b.x.shape[0], 4)

    class OneRank(object):
        def __init__(self, scale):
            self.scale, self.calls = scale, 0

        def all_reduce(self, t):
            assert t.is_cuda and t.shape == (2,)
            t.mul_(self.scale)
            self.calls += 1
    base = gd.p_sample_segment(b, x0, 400, 393, seed=5).cpu().numpy()
    one = OneRank(1.0)
    sharding.enable_global_batch_energy(gd, one)
    same = gd.p_sample_segment(b, x0, 400, 393, seed=5).cpu().numpy()
    assert one.calls == 8 * 3 and np.array_equal(same, base)
    big = OneRank(64.0)
    sharding.enable_global_batch_energy(gd, big)
    other = gd.p_sample_segment(b, x0, 400, 393, seed=5).cpu().numpy()
    assert big.calls == 8 * 3 and not np.array_equal(other, base)
    sharding.enable_global_batch_energy(gd, None)
    assert np.array_equal(gd.p_sample_segment(b, x0, 400, 393, seed=5).cpu().numpy(), base)


@pytest.mark.gpu
def test_mala_native_rccl_allreduce_one_rank(device):
    """ccsp_model_set_energy_allreduce through sharding.enable_global_batch_energy(native): the library itself enqueues
    ncclAllReduce(sum, 2 floats) on the chain's stream between the proposal's energy evaluation and the accept step, on an RCCL
    communicator of its own (ccsp_rccl_unique_id / ccsp_rccl_comm_create; RCCL bound by dlopen).  One rank: the sum over the
    communicator is the shard's own pair, so the chain must be bit-equal to the hook-free one -- and the reduction must really
    run (rejected-step reuse keeps the shard's local energies apart from the reduced pair); removing it restores the plain path."""
    import torch.distributed as dist
    from diffusion_ccsp_amd import ComposedEBMDenoiseFn, ConstraintDiffuser, GaussianDiffusion, sharding
    den = ConstraintDiffuser(dims=worlds.MODE_DIMS['diffuse_pairwise'], hidden_dim=256, input_mode='diffuse_pairwise', EBM='MALA',
                             energy_wrapper=True, device=device, verbose=False)
    den.load_state_dict(weights('weights_diffuse_pairwise_h256_energy.npz'))
    gd = GaussianDiffusion(ComposedEBMDenoiseFn(den), timesteps=1000, EBM='MALA', samples_per_step=3)
    b = worlds.triangular_batch(6, 12, seed=19).to_torch()
    x0 = torch.zeros(b.x.shape[0], 4)
    base = gd.p_sample_segment(b, x0, 400, 380, seed=5).cpu().numpy()
    created = not dist.is_initialized()
    if created:
        import socket
        sk = socket.socket()
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
        sk.close()
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=device)
    try:
        sharding.enable_global_batch_energy(gd, dist)
        assert gd._core()._energy_comm and gd._core()._energy_hook is None
        same = gd.p_sample_segment(b, x0, 400, 380, seed=5).cpu().numpy()
        assert np.array_equal(same, base)
        # the communicator survives a re-creation of the native model (another chain length bound to the same denoiser)
        gd2 = GaussianDiffusion(ComposedEBMDenoiseFn(den), timesteps=200, EBM='MALA', samples_per_step=2)
        gd2.p_sample_segment(b, x0, 100, 95, seed=5)
        assert np.array_equal(gd.p_sample_segment(b, x0, 400, 380, seed=5).cpu().numpy(), base)
        sharding.enable_global_batch_energy(gd, None)
        assert gd._core()._energy_comm is None
        assert np.array_equal(gd.p_sample_segment(b, x0, 400, 380, seed=5).cpu().numpy(), base)
    finally:
        if created:
            dist.destroy_process_group()
