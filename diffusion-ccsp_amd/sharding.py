"""Multi-GPU: independent graph shards, one process per GPU.

The reference has no distributed code at all.  Every operation of the sampling path is per node or
per edge *within one graph* (collation is block-diagonal), so a batch of B graphs shards into
contiguous blocks of graphs with no data-path collective: the chain of each shard runs on its own
GPU.  The only communication is

  * one broadcast of the weights (36.6 MB fp32 at hidden_dim 256) from rank 0 at start-up, and
  * one gather of the final poses ([N, P] fp32, ~37 KB per 256-graph shard) per sampled batch,

both through ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU
tests).  Noise parity across shardings comes from the counter-based stream being indexed by the
*global* node row (noise.py): a shard passes ``row_offset`` = index of its first node.

MALA is the exception: the reference's accept test uses ONE energy for the whole batch
(``logp_x`` / ``logp_x_hat`` of shape [1], ddpm.py:1026-1038), so the shards of a batch are coupled.
Two modes: *replica semantics* (default; each shard is its own reference batch, no communication),
and *global-batch* (``enable_global_batch_energy``): one all-reduce of 2 floats per MALA inner step
(T x S = 10 000 per chain) makes the sharded run the reference's run at the full batch size.
"""
import numpy as np
import torch

from .worlds import GraphBatch


def shard_bounds(n_graphs, rank, world):
    """contiguous block [g0, g1) of graphs owned by `rank` (sizes differ by at most one)"""
    base, rem = divmod(n_graphs, world)
    g0 = rank * base + min(rank, rem)
    return g0, g0 + base + (1 if rank < rem else 0)


def shard_batch(batch, rank, world):
    """sub-batch of a collated numpy GraphBatch for `rank` -> (sub_batch, row_offset).
    `batch.batch` is the graph id of every node (PyG convention)."""
    gid = np.asarray(batch.batch)
    n_graphs = int(gid.max()) + 1 if gid.size else 0
    g0, g1 = shard_bounds(n_graphs, rank, world)
    nodes = np.nonzero((gid >= g0) & (gid < g1))[0]
    if nodes.size == 0:
        r0, r1 = 0, 0
    else:
        r0, r1 = int(nodes[0]), int(nodes[-1]) + 1
    assert nodes.size == r1 - r0, 'graphs must be stored contiguously'
    ei = np.asarray(batch.edge_index)
    sel = (ei[0] >= r0) & (ei[0] < r1)
    assert (((ei[1] >= r0) & (ei[1] < r1)) == sel).all(), 'an edge crosses graphs'
    sub = GraphBatch(x=np.asarray(batch.x)[r0:r1], edge_index=ei[:, sel] - r0,
                     edge_attr=np.asarray(batch.edge_attr)[sel], mask=np.asarray(batch.mask)[r0:r1],
                     batch=gid[r0:r1] - g0, num_graphs=g1 - g0)
    return sub, r0


def broadcast_state_dict(sd, shapes, device, dist=None, src=0):
    """rank `src` holds `sd` (name.weight / name.bias -> array); every rank returns device tensors.
    One flat fp32 buffer, one collective (ring algorithms are per-link bound on xGMI: fewer, larger)."""
    keys = []
    for name, (o, i) in shapes.items():
        keys.append((name + '.weight', (o, i)))
        keys.append((name + '.bias', (o,)))
    total = sum(int(np.prod(s)) for _, s in keys)
    if dist is None or dist.get_world_size() == 1:
        return {k: torch.as_tensor(np.asarray(sd[k]), dtype=torch.float32).to(device) for k, _ in keys}
    flat = torch.empty(total, dtype=torch.float32, device=device)
    if dist.get_rank() == src:
        host = np.concatenate([np.asarray(sd[k], dtype=np.float32).reshape(-1) for k, _ in keys])
        flat.copy_(torch.from_numpy(host))
    dist.broadcast(flat, src=src)
    out, off = {}, 0
    for k, s in keys:
        n = int(np.prod(s))
        out[k] = flat[off:off + n].view(*s).clone()
        off += n
    return out


def gather_poses(x_local, dist=None, sizes=None):
    """final poses of every shard on every rank, concatenated in rank order.  `sizes`: rows per rank
    when shards are uneven (default: equal)."""
    if dist is None or dist.get_world_size() == 1:
        return x_local
    world = dist.get_world_size()
    if sizes is None:
        parts = [torch.empty_like(x_local) for _ in range(world)]
        dist.all_gather(parts, x_local.contiguous())
        return torch.cat(parts, 0)
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
    pad[:x_local.shape[0]] = x_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], 0)


def sample_sharded(sample_fn, batch, dist=None, seed=0):
    """shard `batch` (numpy GraphBatch) by graphs, run `sample_fn(sub_batch, seed, row_offset) -> [n,P]`
    on the local shard, gather.  Every rank returns the full [N, P] result, identical to the
    unsharded run with the same seed."""
    if dist is None or dist.get_world_size() == 1:
        return sample_fn(batch, seed, 0)
    world, rank = dist.get_world_size(), dist.get_rank()
    sub, r0 = shard_batch(batch, rank, world)
    x = sample_fn(sub, seed, r0)
    sizes = []
    for r in range(world):
        s, _ = shard_batch(batch, r, world)
        sizes.append(int(s.x.shape[0]))
    return gather_poses(x, dist, sizes)


class _DevicePair(object):
    """zero-copy torch view of the library's device float[2] {E(x), E(x_hat)}"""

    def __init__(self, ptr):
        self.__cuda_array_interface__ = {'shape': (2,), 'typestr': '<f4', 'data': (int(ptr), False), 'version': 2}


def _native_comm(dist, device):
    """an RCCL communicator of the library's own over the ranks of `dist` (ccsp_rccl_comm_create): rank 0 draws the unique id, the
    process group carries its 128 bytes to the other ranks, every rank joins.  -> ncclComm_t as int"""
    import ctypes as C
    from . import _lib
    L = _lib.lib()
    rank, world = dist.get_rank(), dist.get_world_size()
    buf = C.create_string_buffer(128)
    if rank == 0:
        _lib.check(L.ccsp_rccl_unique_id(buf))
    box = [bytes(buf.raw)]
    comm = C.c_void_p()
    dev = torch.device(device)
    if dev.type == 'cuda':
        # the broadcast too: with the nccl backend broadcast_object_list stages through the CURRENT device, GPU 0 on every rank unless the
        # caller has called set_device
        with torch.cuda.device(dev):
            if world > 1:
                dist.broadcast_object_list(box, src=0, device=dev)
            _lib.check(L.ccsp_rccl_comm_create(world, rank, box[0], C.byref(comm)))
    else:
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        _lib.check(L.ccsp_rccl_comm_create(world, rank, box[0], C.byref(comm)))
    return comm.value


def _group_key(dist):
    """identity of the process group a communicator was made over (the default group of a `torch.distributed`-like module, or the object)"""
    g = getattr(getattr(dist, 'group', None), 'WORLD', None)
    return (id(g if g is not None else dist), dist.get_rank(), dist.get_world_size())


def _destroy_comm(comm):
    try:
        import ctypes as C
        from . import _lib
        _lib.lib().ccsp_rccl_comm_destroy(C.c_void_p(comm))
    except Exception:       # noqa: interpreter shutdown
        pass


def enable_global_batch_energy(gd, dist, native=None):
    """MALA global-batch mode on the HIP path: every inner step's shard energies {E(x), E(x_hat)} are summed over the ranks of `dist`
    on the chain's stream before the accept test (the reference's energies are one scalar for the whole batch, ddpm.py:1026-1038).
    native (default: when the backend is nccl = RCCL): ``ncclAllReduce(sum, 2 floats)`` enqueued by the LIBRARY on a communicator
    of its own (ccsp_model_set_energy_allreduce) -- no Python in the chain.  Otherwise a ctypes hook that calls ``dist.all_reduce``
    (ccsp_model_set_energy_hook; the gloo tests).  Kept on the denoiser and re-installed whenever its native model is
    re-created (weights reloaded, another ``timesteps`` bound).  ``dist=None`` removes it.  The per-timestep acceptance rates a rank
    reads back remain those of its own shard."""
    from . import _lib
    import ctypes as C
    core = gd._core()
    h = gd._handle()
    L = _lib.lib()
    if dist is None:
        _lib.check(L.ccsp_model_set_energy_hook(h, None, None))
        _lib.check(L.ccsp_model_set_energy_allreduce(h, None))
        old = getattr(core, '_energy_comm', None)
        if old:
            fin = getattr(core, '_energy_comm_finalizer', None)
            if fin is not None:
                fin.detach()
            L.ccsp_rccl_comm_destroy(C.c_void_p(old))
        core._energy_hook = None
        core._energy_comm = None
        return
    if native is None:
        native = getattr(dist, 'get_backend', lambda: '')() == 'nccl'
    if native:
        key = _group_key(dist)
        comm = getattr(core, '_energy_comm', None)
        if comm and getattr(core, '_energy_comm_key', None) != key:      # another process group: the cached communicator is not over its ranks
            fin = getattr(core, '_energy_comm_finalizer', None)
            if fin is not None:
                fin.detach()
            L.ccsp_rccl_comm_destroy(C.c_void_p(comm))
            comm = None
        if not comm:
            comm = _native_comm(dist, core.device)
            import weakref
            core._energy_comm_finalizer = weakref.finalize(core, _destroy_comm, comm)      # destroyed with the denoiser
        core._energy_comm = comm
        core._energy_comm_key = key
        core._energy_hook = None
        _lib.check(L.ccsp_model_set_energy_allreduce(h, C.c_void_p(comm)))
        return
    views = {}

    def hook(ctx, ptr, stream):
        try:
            t = views.get(ptr)
            if t is None:
                t = views[ptr] = torch.as_tensor(_DevicePair(ptr), device=core.device)
            dist.all_reduce(t)              # on the current stream = the stream the chain is enqueued on
            return 0
        except Exception:                   # noqa: never let an exception cross the C boundary
            return 1
    cb = _lib.ENERGY_HOOK(hook)
    core._energy_hook = (cb, views)         # keep the trampoline alive as long as the model
    _lib.check(L.ccsp_model_set_energy_hook(h, C.cast(cb, C.c_void_p), None))
