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
    batch = worlds.qualitative_batch(5, 4, seed=8)            # 5 graphs -> uneven shards (3 + 2)
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


def test_world_size_2_gloo_matches_unsharded():
    world, port = 2, _free_port()
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
