"""N > 1 host logic on CPU: world_size-2 gloo processes shard a batch by graphs, run their shard
(with the oracle standing in for the GPU compute: tests may use the checker), gather, and must
reproduce the unsharded result bit for bit (noise rows are global)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import oracle_model, worlds
from diffusion_ccsp_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_sample_fn():
    m = oracle_model('qualitative', 64, 'weights_qualitative_h64.npz', T=1000, S=2)

    def fn(sub, seed, row_offset):
        x = m.graph(sub.to_torch()).chain('ULA', seed=seed, row_offset=row_offset, t_last=985)
        return torch.from_numpy(x)
    return fn


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    batch = worlds.qualitative_batch(5, 4, seed=8)            # 5 graphs -> uneven shards (3 + 2; 2 + 1 + 1 + 1 over four ranks)
    # weights travel from rank 0 only
    from conftest import weights
    W = weights('weights_qualitative_h64.npz') if rank == 0 else None
    shapes = {k[:-7]: v.shape for k, v in weights('weights_qualitative_h64.npz').items() if k.endswith('.weight')}
    sd = sharding.broadcast_state_dict(W, shapes, 'cpu', dist)
    ref = weights('weights_qualitative_h64.npz')
    same = all(np.array_equal(sd[k].numpy(), ref[k]) for k in ref)
    x = sharding.sample_sharded(_oracle_sample_fn(), batch, dist, seed=21)
    q.put((rank, same, x.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_and_batches():
    assert [sharding.shard_bounds(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [sharding.shard_bounds(8, r, 8) for r in range(8)] == [(i, i + 1) for i in range(8)]
    b = worlds.qualitative_batch(5, 4, seed=8)
    s0, r0 = sharding.shard_batch(b, 0, 2)
    s1, r1 = sharding.shard_batch(b, 1, 2)
    assert r0 == 0 and r1 == 15 and s0.x.shape[0] == 15 and s1.x.shape[0] == 10
    assert s0.edge_index.shape[1] + s1.edge_index.shape[1] == b.edge_index.shape[1]
    assert s1.edge_index.min() >= 0 and s1.edge_index.max() < 10
    assert np.array_equal(np.concatenate([s0.x, s1.x]), b.x)


import pytest


@pytest.mark.parametrize('world', [2, 4])
def test_world_size_n_gloo_matches_unsharded(world):
    """uneven shards (5 graphs over 2 or 4 ranks): the padded all_gather of gather_poses(sizes=...) puts every shard's rows back in place"""
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    batch = worlds.qualitative_batch(5, 4, seed=8)
    want = _oracle_sample_fn()(batch, 21, 0).numpy()
    for rank, same, x in got:
        assert same, 'broadcast weights differ on rank %d' % rank
        assert np.array_equal(x, want), rank


# ---------------------------------------------------------------- MALA: the batch-scalar energies couple the shards

def _mala_model():
    return oracle_model('diffuse_pairwise', 64, 'weights_diffuse_pairwise_h64_energy.npz', T=1000, S=3, energy=True)


def _mala_batch():
    return worlds.triangular_batch(3, 6, seed=19)             # 3 graphs -> shards of 2 + 1


def _mala_start():
    b = _mala_batch()
    x0 = (np.random.default_rng(4).standard_normal((b.x.shape[0], 4)) * 0.3).astype(np.float32)
    m = b.mask.astype(bool)
    x0[m] = b.x[m][:, 3:7]
    return x0


def _mala_worker(rank, world, port, q, global_batch):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    m = _mala_model()
    calls = [0]
    if global_batch:
        def hook(pair):                                       # {E(x), E(x_hat)} of this shard -> of the whole batch
            t = torch.from_numpy(pair)
            dist.all_reduce(t)
            calls[0] += 1
        m.set_energy_hook(hook)

    x0 = _mala_start()

    def fn(sub, seed, row_offset):
        xs = x0[row_offset:row_offset + sub.x.shape[0]]
        return torch.from_numpy(m.graph(sub.to_torch()).chain('MALA', seed=seed, row_offset=row_offset, x=xs, t_first=300, t_last=293))
    x = sharding.sample_sharded(fn, _mala_batch(), dist, seed=23)
    q.put((rank, x.numpy(), calls[0]))
    dist.barrier()
    dist.destroy_process_group()


def _run_mala(global_batch, world=2):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_mala_worker, args=(r, world, port, q, global_batch)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_mala_global_batch_world_size_2_matches_unsharded():
    """with the shards' energies summed before every accept test (one 2-float all_reduce per MALA inner step)
    the sharded chain IS the reference's chain at the full batch size: bit-equal to the unsharded oracle run.  Without the
    hook each shard is its own reference batch (replica semantics): a different, equally valid, chain."""
    batch = _mala_batch()
    x0 = _mala_start()
    want, acc = _mala_model().graph(batch.to_torch()).chain('MALA', seed=23, x=x0, t_first=300, t_last=293, accept=True)
    assert 0.05 < acc[293:301].mean() < 0.95                  # timesteps where proposals are both accepted and rejected
    got = _run_mala(True)
    for rank, x, calls in got:
        assert calls == 8 * 3, calls                          # one reduction per inner step of the 8 timesteps
        assert np.array_equal(x, want), rank
    replica = _run_mala(False)
    assert np.array_equal(replica[0][1], replica[1][1]) and not np.array_equal(replica[0][1], want)
    # replica semantics = every shard run as a batch of its own
    parts = []
    for r in range(2):
        sub, r0 = sharding.shard_batch(batch, r, 2)
        parts.append(_mala_model().graph(sub.to_torch()).chain('MALA', seed=23, row_offset=r0, x=x0[r0:r0 + sub.x.shape[0]], t_first=300, t_last=293))
    assert np.array_equal(replica[0][1], np.concatenate(parts))


def test_mala_global_batch_world_size_4_uneven_shards():
    """four ranks, five graphs (2 + 1 + 1 + 1): the all-reduced energies make every rank's shard follow the unsharded chain"""
    b = worlds.triangular_batch(5, 6, seed=19)
    x0 = (np.random.default_rng(4).standard_normal((b.x.shape[0], 4)) * 0.3).astype(np.float32)
    mk = b.mask.astype(bool)
    x0[mk] = b.x[mk][:, 3:7]
    want = _mala_model().graph(b.to_torch()).chain('MALA', seed=23, x=x0, t_first=300, t_last=296)
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_mala_worker4, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    got = [q.get(timeout=900) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, x, calls in got:
        assert calls == 5 * 3 and np.array_equal(x, want), rank


def _mala_worker4(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    m = _mala_model()
    calls = [0]

    def hook(pair):
        t = torch.from_numpy(pair)
        dist.all_reduce(t)
        calls[0] += 1
    m.set_energy_hook(hook)
    b = worlds.triangular_batch(5, 6, seed=19)
    x0 = (np.random.default_rng(4).standard_normal((b.x.shape[0], 4)) * 0.3).astype(np.float32)
    mk = b.mask.astype(bool)
    x0[mk] = b.x[mk][:, 3:7]

    def fn(sub, seed, row_offset):
        xs = x0[row_offset:row_offset + sub.x.shape[0]]
        return torch.from_numpy(m.graph(sub.to_torch()).chain('MALA', seed=seed, row_offset=row_offset, x=xs, t_first=300, t_last=296))
    x = sharding.sample_sharded(fn, b, dist, seed=23)
    q.put((rank, x.numpy(), calls[0]))
    dist.barrier()
    dist.destroy_process_group()
