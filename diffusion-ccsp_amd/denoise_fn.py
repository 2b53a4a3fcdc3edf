"""Host-side mirror of the reference's denoiser interface (networks/denoise_fn.py) on top of the
HIP library.  Same class names, constructor arguments, attributes and call signatures as the
reference so that ``GaussianDiffusion`` / ``Trainer.evaluate`` / ``visualize_energy.py`` style
callers work unchanged; the arithmetic happens in csrc/ccsp_hip.hip.

    ConstraintDiffuser(dims, hidden_dim, ..., input_mode, EBM, normalize, energy_wrapper, device)
        reference: networks/denoise_fn.py:184-291 (ctor), :453-548 (forward)
    ComposedEBMDenoiseFn(model)      reference: networks/denoise_fn.py:57-83

PyTorch is used for device memory and streams only.
"""
import ctypes as C
import math
import weakref

import numpy as np
import torch

from . import _lib
from .worlds import constraint_set


MODEL_KINDS = {'Diffusion-CCSP': 0, 'StructDiffusion': 1}


def param_names(n_types, grasp, model='Diffusion-CCSP'):
    """(weight key, bias key) pairs in reference state_dict order (networks/denoise_fn.py:227-308; the
    StructDiffusion tail follows networks/transformer.py:43-57)"""
    names = ['geom_encoder.0', 'geom_encoder.2']
    if grasp:
        names += ['grasp_encoder.0', 'grasp_encoder.2']
    names += ['pose_encoder.0', 'pose_encoder.2', 'pose_decoder.0', 'pose_decoder.2', 'time_mlp.1', 'time_mlp.3']
    if model != 'StructDiffusion':
        names += ['mlps.%d.0' % i for i in range(n_types)]
        return [(n + '.weight', n + '.bias') for n in names]
    out = [(n + '.weight', n + '.bias') for n in names]
    out.append(('ln_pre.weight', 'ln_pre.bias'))
    for l in range(4):
        pre = 'transformer.resblocks.%d.' % l
        out += [(pre + 'attn.in_proj_weight', pre + 'attn.in_proj_bias')]
        out += [(pre + n + '.weight', pre + n + '.bias') for n in ('attn.out_proj', 'ln_1', 'mlp.c_fc', 'mlp.c_proj', 'ln_2')]
    out.append(('ln_post.weight', 'ln_post.bias'))
    return out


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


class _SubBatch(object):
    """one domain's share of a composed batch (compose()): the four tensors _Graph reads"""

    def __init__(self, x, edge_index, edge_attr, mask):
        self.x, self.edge_index, self.edge_attr, self.mask = x, edge_index.contiguous(), edge_attr.contiguous(), mask


class _Graph(object):
    """one collated batch uploaded to the library (ccsp_graph_create)"""

    def __init__(self, owner, batch):
        dev = owner.device
        self.owner = owner
        self.x = batch.x.detach().to(dev, torch.float32).contiguous()
        self.edge_index = batch.edge_index.detach().to(dev, torch.int64).contiguous()
        self.edge_attr = batch.edge_attr.detach().to(dev, torch.float32).contiguous()
        self.mask = batch.mask.detach().to(dev, torch.int8).contiguous()
        self.N, self.F = self.x.shape
        self.E = self.edge_index.shape[1] if self.edge_index.dim() == 2 else 0
        h = C.c_void_p()
        L = _lib.lib()
        _lib.check(L.ccsp_graph_create(owner._handle(), self.N, self.E, self.F, _ptr(self.x), _ptr(self.edge_index),
                                       _ptr(self.edge_attr), _ptr(self.mask), _stream_ptr(dev), C.byref(h)))
        self.h = h
        self.model_handle = owner._generation       # (a new native model may reuse a freed one's address)
        owner._live_graphs.add(self)
        if owner.model == 'StructDiffusion':
            # the token sequences: batch.batch, and batch.shuffled when the dataset carries it (denoise_fn.py:408-417)
            self.seq = batch.batch.detach().to(dev, torch.int64).contiguous()
            sh = getattr(batch, 'shuffled', None)
            self.shuffled = None if sh is None else sh.detach().to(dev, torch.int64).contiguous()
            _lib.check(L.ccsp_graph_set_sequences(h, _ptr(self.seq), None if self.shuffled is None else _ptr(self.shuffled),
                                                  _stream_ptr(dev)))

    def destroy(self):
        if getattr(self, 'h', None):
            _lib.lib().ccsp_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class ConstraintDiffuser(object):
    """drop-in for the reference class of the same name; model='Diffusion-CCSP' (the paper's method) or
    'StructDiffusion' (the transformer baseline, denoise_fn.py:267-282,391-451)."""

    def __init__(self, dims=((2, 0, 2), (2, 2, 4)), hidden_dim=256, max_num_obj=12, input_mode=None,
                 EBM=False, pretrained=False, normalize=True, energy_wrapper=False, device='cuda',
                 model='Diffusion-CCSP', verbose=True, timesteps=1000):
        if model not in MODEL_KINDS:
            raise NotImplementedError("model=%r: 'Diffusion-CCSP' and 'StructDiffusion' are built" % model)
        if model == 'StructDiffusion' and energy_wrapper:
            raise ValueError('StructDiffusion has no energy mode')
        if input_mode is None:
            raise ValueError('input_mode is required')
        self.hidden_dim = hidden_dim
        self.max_num_obj = max_num_obj
        self.EBM = EBM
        self.device = torch.device(device)
        self.dims = tuple(tuple(d) for d in dims)
        self.input_mode = input_mode
        self.use_image = False
        self.normalize = normalize
        self.verbose = verbose
        self.energy_wrapper = energy_wrapper
        self.model = model
        self.constraint_sets = constraint_set(input_mode)
        self.ebm_per_steps = 1
        self.training = False
        self.timesteps = timesteps
        self._grasp = 'robot' in input_mode
        if self._grasp and len(self.dims) != 3:
            raise ValueError("'robot' input modes need dims with a grasp group")
        # composing with a second domain (denoise_fn.py:287-291): set by compose()
        self.pose_encoder_2 = self.geom_encoder_2 = self.pose_decoder_2 = self.time_mlp_2 = None
        self.composing_weight = (1, 1)
        self._second = None       # the second domain's ConstraintDiffuser
        self._params = None       # name -> device tensor
        self._h = None
        self._energy_hook = None  # (ctypes trampoline, views) of the MALA shard hook: re-installed on every new native model
        self._energy_comm = None  # ncclComm_t of the in-library reduction (sharding.enable_global_batch_energy, native form)
        self._generation = 0      # bumped for every native model created: handles are compared by this, not by address
        self._graphs = {}                     # id(batch) -> (weakref to the batch, content key, _Graph)
        self._live_graphs = weakref.WeakSet()  # every _Graph built on the current native model, wherever it is referenced
        if self.device.type != 'cuda':
            raise _lib.CcspError("ConstraintDiffuser(device=%r): the HIP path needs a GPU device ('cuda'); "
                                 "there is no CPU fallback" % (device,))

    def _n_own_types(self):
        """constraint types whose MLPs live in THIS model (a composed model lists the second domain's types after them)"""
        return self._n_first if self._second is not None else len(self.constraint_sets)

    # ---- weights ------------------------------------------------------------------------
    def shapes(self):
        H, P = self.hidden_dim, self.dims[-1][0]
        kin = H * (6 if self._grasp else 5)
        sh = {'geom_encoder.0': (H // 2, self.dims[0][0]), 'geom_encoder.2': (H, H // 2),
              'pose_encoder.0': (H // 2, P), 'pose_encoder.2': (H, H // 2),
              'pose_decoder.0': (H // 2, H), 'pose_decoder.2': (P, H // 2),
              'time_mlp.1': (4 * H, H), 'time_mlp.3': (H, 4 * H)}
        if self._grasp:
            sh.update({'grasp_encoder.0': (H // 2, self.dims[1][0]), 'grasp_encoder.2': (H, H // 2)})
        if self.model == 'StructDiffusion':
            W = H * (3 if self._grasp else 2)
            sh.update({'ln_pre': (W,), 'ln_post': (W,)})
            for l in range(4):
                pre = 'transformer.resblocks.%d.' % l
                sh.update({pre + 'attn.in_proj_': (3 * W, W), pre + 'attn.out_proj': (W, W), pre + 'ln_1': (W,),
                           pre + 'mlp.c_fc': (4 * W, W), pre + 'mlp.c_proj': (W, 4 * W), pre + 'ln_2': (W,)})
            return sh
        for i in range(self._n_own_types()):
            sh['mlps.%d.0' % i] = (2 * H, kin)
        return sh

    @staticmethod
    def _keys(name):
        """state_dict keys of one module: nn.MultiheadAttention stores in_proj_weight / in_proj_bias"""
        if name.endswith('in_proj_'):
            return name + 'weight', name + 'bias'
        return name + '.weight', name + '.bias'

    def load_state_dict(self, sd, strict=True):
        """accepts the reference's key names, bare or with the 'denoise_fn.' / 'denoise_fn.model.' prefixes
        a GaussianDiffusion checkpoint carries (networks/ddpm.py:503-514)"""
        clean = {}
        for k, v in sd.items():
            for pre in ('denoise_fn.model.', 'denoise_fn.', 'model.'):
                if k.startswith(pre):
                    k = k[len(pre):]
                    break
            clean[k] = v
        params = {}
        for name, shape in self.shapes().items():
            wk, bk = self._keys(name)
            for key, shp in ((wk, shape), (bk, (shape[0],))):
                if key not in clean:
                    raise KeyError('missing key %s in state_dict' % key)
                t = clean[key]
                if isinstance(t, np.ndarray):
                    t = torch.from_numpy(np.ascontiguousarray(t))
                t = t.detach().to(self.device, torch.float32).contiguous()
                if tuple(t.shape) != tuple(shp):
                    raise ValueError('size mismatch for %s: %s vs %s' % (key, tuple(t.shape), tuple(shp)))
                params[key] = t
        self._params = params
        self._drop_handle()
        return self

    def reset_parameters(self, seed=0):
        """nn.Linear default init (uniform +-1/sqrt(fan_in)), for plumbing runs without a checkpoint"""
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for name, shape in self.shapes().items():
            wk, bk = self._keys(name)
            if len(shape) == 1:                         # nn.LayerNorm
                sd[wk], sd[bk] = torch.ones(shape), torch.zeros(shape)
                continue
            o, i = shape
            bound = 1.0 / math.sqrt(i)
            sd[wk] = (torch.rand((o, i), generator=g) * 2 - 1) * bound
            sd[bk] = (torch.rand((o,), generator=g) * 2 - 1) * bound
        return self.load_state_dict(sd)

    def state_dict(self):
        if self._params is None:
            raise _lib.CcspError('no weights loaded')
        return dict(self._params)

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError('training is outside the sampling path (SURVEY 2, row 1)')
        return self

    def to(self, *a, **k):
        return self

    def cuda(self):
        return self

    # ---- composition of two domains -----------------------------------------------------
    def compose(self, second, composing_weight=(1, 1)):
        """Attach a second constraint domain the way the reference's 'robot_qualitative' mode expects it (denoise_fn.py:
        287-291,310-311): `second` is the other domain's ConstraintDiffuser (qualitative: pose_dim = this one's - 1, its
        poses are [x, y | the last columns of batch.x], denoise_fn.py:499).  Afterwards constraint types >=
        len(self.constraint_sets) are the second domain's types (renumbered from there), its encoders / decoder / time MLP
        are reachable as pose_encoder_2 / geom_encoder_2 / pose_decoder_2 / time_mlp_2, and forward / p_sample_loop evaluate
        both domains (ccsp_compose_denoise / ccsp_compose_chain_run).  composing_weight as in the reference (:291,362-370)."""
        if self.input_mode != 'robot_qualitative':
            raise ValueError("compose(): the reference composes domains under input_mode 'robot_qualitative' (denoise_fn.py:311)")
        if self.model != 'Diffusion-CCSP' or second.model != 'Diffusion-CCSP':
            raise NotImplementedError('compose(): Diffusion-CCSP models only')
        if second.dims[-1][0] + 1 != self.dims[-1][0] or second.hidden_dim != self.hidden_dim:
            raise ValueError('compose(): the second domain needs pose_dim %d and hidden_dim %d' % (self.dims[-1][0] - 1, self.hidden_dim))
        if second._params is None or self._params is None:
            raise _lib.CcspError('compose(): load the weights of both models first')
        self._n_first = len(constraint_set(self.input_mode))
        self.constraint_sets = list(constraint_set(self.input_mode)) + list(second.constraint_sets)
        self._second = second
        self.composing_weight = tuple(composing_weight)
        self.pose_encoder_2, self.geom_encoder_2 = second.pose_encoder, second.geom_encoder
        self.time_mlp_2 = second.time_mlp
        self.pose_decoder_2 = second          # (the reference's attribute holds the module; the decoder runs inside _process_constraint)
        return self

    def _composed_parts(self):
        """the two single-domain models of the composed forward, bound to the same number of timesteps and to the same mode (the
        second native model follows this one's energy_wrapper: forward() and GaussianDiffusion._run() both come through here)"""
        if self._second.energy_wrapper != self.energy_wrapper:
            self._second.energy_wrapper = self.energy_wrapper
            self._second._drop_handle()
        self._second._bind(self.timesteps)
        self._handle()
        self._second._handle()
        return self, self._second

    def _compose_struct(self):
        return _lib.Compose(zero_col=2, weight_first=float(self.composing_weight[0]), weight_second=float(self.composing_weight[1]),
                            normalize=int(bool(self.normalize)))

    def _composed_graphs(self, batch):
        """(first-domain _Graph, second-domain _Graph) of a batch whose edge types run over both domains; cached like _graph()"""
        first, second = self._composed_parts()
        fields = (batch.x, batch.edge_index, batch.edge_attr, batch.mask)
        key = tuple((t.data_ptr(), tuple(t.shape), t._version) for t in fields)
        ent = self._graphs.get(id(batch))
        if ent is not None:
            ref, k, gs = ent
            if ref() is batch and k == key and isinstance(gs, tuple) and gs[0].h and gs[1].h and \
                    gs[0].model_handle == first._generation and gs[1].model_handle == second._generation:
                return gs
            del self._graphs[id(batch)]
        n1 = self._n_first
        ea = batch.edge_attr.detach().to(torch.float32)
        ei = batch.edge_index.detach()
        sel1, sel2 = ea < n1, ea >= n1
        x = batch.x.detach().to(torch.float32)
        g2w, p2 = self._second.dims[0][0], self._second.dims[-1][0]
        x2 = torch.cat([x[:, self.dims[0][1]:self.dims[0][1] + g2w], torch.zeros((x.shape[0], p2), dtype=x.dtype, device=x.device)], dim=1)
        b1 = _SubBatch(x, ei[:, sel1], ea[sel1], batch.mask)
        b2 = _SubBatch(x2, ei[:, sel2], ea[sel2] - n1, batch.mask)
        with torch.cuda.device(self.device):
            gs = (first._graph(b1), second._graph(b2))
        gs[0]._keep, gs[1]._keep = b1, b2
        bid = id(batch)
        try:
            ref = weakref.ref(batch, lambda _r, d=self._graphs, i=bid: d.pop(i, None))
        except TypeError:
            return gs
        while len(self._graphs) > 8:
            self._graphs.pop(next(iter(self._graphs)))
        self._graphs[bid] = (ref, key, gs)
        return gs

    # ---- native handles -----------------------------------------------------------------
    def _drop_handle(self):
        # graphs first, then the model they were built on (a GaussianDiffusion may still hold one as _last_graph; the
        # library also tolerates the other order: ccsp_model_destroy orphans the graphs it leaves behind)
        for g in list(self._live_graphs):
            g.destroy()
        self._live_graphs = weakref.WeakSet()
        self._graphs.clear()
        if self._h is not None:
            _lib.lib().ccsp_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._drop_handle()
        except Exception:
            pass

    def _bind(self, timesteps):
        if timesteps != self.timesteps:
            self.timesteps = timesteps
            self._drop_handle()

    def _handle(self):
        if self._h is not None:
            return self._h
        if self._params is None:
            raise _lib.CcspError('ConstraintDiffuser has no weights: call load_state_dict() (or reset_parameters())')
        L = _lib.lib()
        grasp = self._grasp
        d = _lib.ModelDesc(hidden_dim=self.hidden_dim, pose_dim=self.dims[-1][0], pose_begin=self.dims[-1][1],
                           geom_dim=self.dims[0][0], grasp_dim=self.dims[1][0] if grasp else 0,
                           grasp_begin=self.dims[1][1] if grasp else 0, n_types=self._n_own_types(),
                           timesteps=self.timesteps, normalize=int(bool(self.normalize)),
                           energy_wrapper=int(bool(self.energy_wrapper)), ebm_per_steps=int(self.ebm_per_steps),
                           model_kind=MODEL_KINDS[self.model])
        ptrs = []
        for wk, bk in param_names(self._n_own_types(), grasp, self.model):
            ptrs.append(self._params[wk].data_ptr())
            ptrs.append(self._params[bk].data_ptr())
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.ccsp_model_create(C.byref(d), arr, _stream_ptr(self.device), C.byref(h)))
        self._h = h
        self._generation += 1
        hook = getattr(self, '_energy_hook', None)
        if hook is not None:
            # MALA global-batch mode (sharding.enable_global_batch_energy): the hook lives on the native model, which was just
            # re-created (weights reloaded, or another `timesteps` bound) -- without it the shards would silently decouple
            _lib.check(L.ccsp_model_set_energy_hook(h, C.cast(hook[0], C.c_void_p), None))
        comm = getattr(self, '_energy_comm', None)
        if comm:                                  # the native form of the same coupling (ccsp_model_set_energy_allreduce)
            _lib.check(L.ccsp_model_set_energy_allreduce(h, C.c_void_p(comm)))
        return h

    def _graph(self, batch):
        """graph handle for this batch (static over a chain), cached per batch OBJECT: the entry dies with the
        batch (weak reference), and is rebuilt when one of its four tensors was replaced or edited in place
        (storage pointer, shape and torch's in-place version counter of x / edge_index / edge_attr / mask)"""
        fields = (batch.x, batch.edge_index, batch.edge_attr, batch.mask)
        key = tuple((t.data_ptr(), tuple(t.shape), t._version) for t in fields)
        self._handle()
        ent = self._graphs.get(id(batch))
        if ent is not None:
            ref, k, g = ent
            if ref() is batch and k == key and g.h and g.model_handle == self._generation:
                return g
            del self._graphs[id(batch)]
        with torch.cuda.device(self.device):
            g = _Graph(self, batch)
        bid = id(batch)
        try:
            ref = weakref.ref(batch, lambda _r, d=self._graphs, i=bid: d.pop(i, None))
        except TypeError:                         # a batch type without weak-reference support: not cached
            return g
        while len(self._graphs) > 8:
            self._graphs.pop(next(iter(self._graphs)))
        self._graphs[bid] = (ref, key, g)
        return g

    # ---- the reference's call surface ---------------------------------------------------
    def time_mlp(self, t):
        """time embedding rows for timestep values t [n] (integer or float, as visualize_energy.py:409 passes them) -> [n, H]
        (SinusoidalPosEmb + the time MLP, denoise_fn.py:38-50,259-264)"""
        tv = torch.as_tensor(t).detach().to(self.device, torch.float32).reshape(-1).contiguous()
        out = torch.empty((tv.shape[0], self.hidden_dim), device=self.device, dtype=torch.float32)
        if tv.shape[0] == 0:
            return out
        _lib.check(_lib.lib().ccsp_time_mlp(self._handle(), tv.shape[0], _ptr(tv), _ptr(out), _stream_ptr(self.device)))
        return out

    def _encode(self, which, x, in_dim):
        x = torch.as_tensor(x).detach().to(self.device, torch.float32)
        if x.shape[-1] != in_dim:
            raise ValueError('encoder input has %d columns, expected %d' % (x.shape[-1], in_dim))
        flat = x.reshape(-1, in_dim).contiguous()
        out = torch.empty((flat.shape[0], self.hidden_dim), device=self.device, dtype=torch.float32)
        if flat.shape[0]:
            _lib.check(_lib.lib().ccsp_encode(self._handle(), which, flat.shape[0], _ptr(flat), _ptr(out), _stream_ptr(self.device)))
        return out.reshape(tuple(x.shape[:-1]) + (self.hidden_dim,))

    # the denoiser's sub-modules as callables on the caller's own tensors, any leading shape (visualize_energy.py:402-450)
    def geom_encoder(self, x):
        """denoise_fn.geom_encoder (denoise_fn.py:227-236): [..., dims[0][0]] -> [..., H]"""
        return self._encode(0, x, self.dims[0][0])

    def pose_encoder(self, x):
        """denoise_fn.pose_encoder (denoise_fn.py:241-250): [..., P] -> [..., H]"""
        return self._encode(1, x, self.dims[-1][0])

    def grasp_encoder(self, x):
        if not self._grasp:
            raise AttributeError("grasp_encoder exists for 'robot' input modes only")
        return self._encode(2, x, self.dims[1][0])

    def _process_constraint(self, i, input_dict):
        """ConstraintDiffuser._process_constraint (denoise_fn.py:341-371): the type-i MLP on [geoms_emb | poses_emb |
        time_embedding] (+ grasp_emb first for 'robot' modes) and the pose decoder on both output halves -> [b, 2, P]"""
        if self._second is not None and i >= self._n_first:
            # second-domain type (denoise_fn.py:342-344,364-370): its own MLP and decoder, a zero column at index 2, its weight
            d = {'geoms_emb': input_dict['geoms_emb_2'], 'poses_emb': input_dict['poses_emb_2'], 'time_embedding': input_dict['time_embedding']}
            o = self._second._process_constraint(i - self._n_first, d)
            o = torch.cat([o[:, :, :2], torch.zeros_like(o[:, :, 0:1]), o[:, :, 2:]], dim=-1)
            return o * self.composing_weight[1] if self.composing_weight[1] != 1 else o
        out = self._process_own_constraint(i, input_dict)
        return out * self.composing_weight[0] if self.composing_weight[0] != 1 else out

    def _process_own_constraint(self, i, input_dict):
        ge = input_dict['geoms_emb'].detach().to(self.device, torch.float32).contiguous()
        pe = input_dict['poses_emb'].detach().to(self.device, torch.float32).contiguous()
        te = input_dict['time_embedding'].detach().to(self.device, torch.float32).contiguous()
        b, H = ge.shape[0], self.hidden_dim
        if tuple(ge.shape) != (b, 2, H) or tuple(pe.shape) != (b, 2, H) or tuple(te.shape) != (b, H):
            raise ValueError('_process_constraint: geoms_emb / poses_emb must be [b, 2, %d] and time_embedding [b, %d]' % (H, H))
        gr = None
        if self._grasp:
            gr = input_dict['grasp_emb'].detach().to(self.device, torch.float32).contiguous()
        out = torch.empty((b, 2, self.dims[-1][0]), device=self.device, dtype=torch.float32)
        if b:
            _lib.check(_lib.lib().ccsp_process_constraint(self._handle(), int(i), b, _ptr(ge), _ptr(pe), _ptr(te),
                                                          None if gr is None else _ptr(gr), _ptr(out), _stream_ptr(self.device)))
        return out

    def edge_outputs(self, poses_in, batch, t):
        """_process_constraint outputs of every edge, [E, 2, P] in the caller's edge order"""
        g = self._graph(batch)
        p = poses_in.detach().to(self.device, torch.float32).contiguous()
        out = torch.empty((g.E, 2, self.dims[-1][0]), device=self.device, dtype=torch.float32)
        _lib.check(_lib.lib().ccsp_edge_outputs(self._h, g.h, _ptr(p), int(torch.as_tensor(t).reshape(-1)[0]),
                                                _ptr(out), _stream_ptr(self.device)))
        return out

    def forward(self, poses_in, batch, t, verbose=False, debug=False, tag='EBM', eval=False):
        """ConstraintDiffuser.forward (denoise_fn.py:453-537).  direct mode -> [N,P];
        energy mode (tag == 'EBM' and energy_wrapper) -> (gradients [N,P], energy scalar)"""
        L = _lib.lib()
        if self._second is not None:
            energy_mode = tag == 'EBM' and self.energy_wrapper
            if energy_mode and tuple(self.composing_weight) != (1, 1):
                raise NotImplementedError('the energy of composed domains is built for composing_weight (1, 1)')
            g1, g2 = self._composed_graphs(batch)
            first, second = self, self._second
            p = poses_in.detach().to(self.device, torch.float32).contiguous()
            out = torch.empty_like(p)
            c = self._compose_struct()
            if energy_mode:
                energy = torch.zeros((), device=self.device, dtype=torch.float32)
                _lib.check(L.ccsp_compose_energy_grad(first._h, g1.h, second._h, g2.h, C.byref(c), _ptr(p), int(torch.as_tensor(t).reshape(-1)[0]),
                                                      _ptr(out), _ptr(energy), _stream_ptr(self.device)))
                return out, energy
            _lib.check(L.ccsp_compose_denoise(first._h, g1.h, second._h, g2.h, C.byref(c), _ptr(p), int(torch.as_tensor(t).reshape(-1)[0]),
                                              _ptr(out), _stream_ptr(self.device)))
            return out
        g = self._graph(batch)
        p = poses_in.detach().to(self.device, torch.float32).contiguous()
        tv = int(torch.as_tensor(t).reshape(-1)[0])
        out = torch.empty_like(p)
        if tag == 'EBM' and self.energy_wrapper:
            energy = torch.zeros((), device=self.device, dtype=torch.float32)
            _lib.check(L.ccsp_energy_grad(self._h, g.h, _ptr(p), tv, _ptr(out), _ptr(energy), _stream_ptr(self.device)))
            return out, energy
        _lib.check(L.ccsp_denoise(self._h, g.h, _ptr(p), tv, _ptr(out), _stream_ptr(self.device)))
        return out

    __call__ = forward


class ComposedEBMDenoiseFn(object):
    """wrapper exposing forward -> gradients and neg_logp_unnorm -> energy
    (reference networks/denoise_fn.py:57-83)"""

    def __init__(self, model, ebm_per_steps=1):
        self.model = model
        self.device = model.device
        self.dims = model.dims
        self.input_mode = model.input_mode
        self.ebm_per_steps = ebm_per_steps
        self.energy_wrapper = True
        model.energy_wrapper = True
        model.ebm_per_steps = ebm_per_steps
        model._drop_handle()
        self.training = False

    def neg_logp_unnorm(self, poses_in, batch, t, **kwargs):
        kwargs['tag'] = 'EBM'
        gradients, energy = self.model.forward(poses_in, batch, t, **kwargs)
        return energy.sum()

    def forward(self, poses_in, batch, t, **kwargs):
        if isinstance(poses_in, np.ndarray):
            poses_in = torch.tensor(poses_in, device=self.model.device)
            t = torch.tensor(t, device=self.model.device)
        kwargs['tag'] = 'EBM'
        gradients, energy = self.model.forward(poses_in, batch, t, **kwargs)
        return gradients

    __call__ = forward

    def eval(self):
        return self

    def load_state_dict(self, sd, strict=True):
        self.model.load_state_dict(sd, strict)
        return self

    def state_dict(self):
        return {'model.' + k: v for k, v in self.model.state_dict().items()}
