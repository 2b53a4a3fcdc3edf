"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle (oracle/ccsp_oracle.c).

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the *checker*.
The product package (diffusion-ccsp_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

SAMPLERS = {False: 0, None: 0, 'NONE': 0, 'ULA': 1, 'ULA+': 2, 'MALA': 3, 'HMC': 4}
SCHEDULE_KEYS = ['betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_recip_alphas_cumprod',
                 'sqrt_recipm1_alphas_cumprod', 'posterior_log_variance_clipped', 'posterior_mean_coef1',
                 'posterior_mean_coef2', 'kappa', 'step_sizes', 'posterior_variance']


class Desc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'hidden_dim', 'pose_dim', 'pose_begin', 'geom_dim', 'grasp_dim', 'grasp_begin', 'n_types',
        'timesteps', 'normalize', 'energy_wrapper', 'ebm_per_steps', 'model_kind')]


class Noise(C.Structure):
    _fields_ = [('mode', C.c_int32), ('_pad', C.c_int32), ('seed', C.c_uint64), ('row_offset', C.c_uint64),
                ('normal', C.c_void_p), ('n_normal', C.c_uint64), ('uniform', C.c_void_p),
                ('n_uniform', C.c_uint64), ('call_base', C.c_uint64), ('ucall_base', C.c_uint64)]


def build(force=False):
    """compile both oracle variants with gcc (idempotent)"""
    outs = [os.path.join(HERE, 'libccsp_oracle.so'), os.path.join(HERE, 'libccsp_oracle_f64.so')]
    src = os.path.join(HERE, 'ccsp_oracle.c')
    if force or any((not os.path.isfile(o)) or os.path.getmtime(o) < os.path.getmtime(src) for o in outs):
        subprocess.check_call(['make', '-s', '-C', HERE, '-B'] if force else ['make', '-s', '-C', HERE])
    return outs


_libs = {}


def lib(f64=False):
    key = bool(f64)
    if key in _libs:
        return _libs[key]
    path = os.path.join(HERE, 'libccsp_oracle_f64.so' if f64 else 'libccsp_oracle.so')
    if not os.path.isfile(path):
        build()
    try:
        L = C.CDLL(path)
    except OSError:
        build(force=True)
        L = C.CDLL(path)
    L.ccspo_last_error.restype = C.c_char_p
    vp = C.c_void_p
    L.ccspo_model_create.argtypes = [C.POINTER(Desc), C.POINTER(vp), C.POINTER(vp)]
    L.ccspo_model_destroy.argtypes = [vp]
    L.ccspo_model_destroy.restype = None
    L.ccspo_schedule_set.argtypes = [vp, vp, vp, vp, C.c_int32]
    L.ccspo_schedule_get.argtypes = [vp, C.c_int32, vp]
    L.ccspo_time_embedding.argtypes = [vp, C.c_int32, vp]
    L.ccspo_graph_create.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, C.POINTER(vp)]
    L.ccspo_graph_set_sequences.argtypes = [vp, vp, vp]
    L.ccspo_graph_destroy.argtypes = [vp]
    L.ccspo_graph_destroy.restype = None
    L.ccspo_denoise.argtypes = [vp, vp, vp, C.c_int32, vp]
    L.ccspo_energy_grad.argtypes = [vp, vp, vp, C.c_int32, vp, vp]
    L.ccspo_energy_grad_split.argtypes = [vp, vp, vp, vp, C.c_int32, C.c_int32, vp, vp]
    L.ccspo_edge_outputs.argtypes = [vp, vp, vp, C.c_int32, vp]
    L.ccspo_chain_run.argtypes = [vp, vp, C.c_int32, C.POINTER(Noise), vp, C.c_int32, C.c_int32, C.c_int32, vp, vp]
    _libs[key] = L
    return L


def param_order(n_types, grasp, struct_diffusion=False):
    """(module name, weight key suffix, bias key suffix) in the order the C side expects"""
    names = ['geom_encoder.0', 'geom_encoder.2']
    if grasp:
        names += ['grasp_encoder.0', 'grasp_encoder.2']
    names += ['pose_encoder.0', 'pose_encoder.2', 'pose_decoder.0', 'pose_decoder.2', 'time_mlp.1', 'time_mlp.3']
    out = [(n, '.weight', '.bias') for n in names]
    if not struct_diffusion:
        return out + [('mlps.%d.0' % i, '.weight', '.bias') for i in range(n_types)]
    out.append(('ln_pre', '.weight', '.bias'))
    for l in range(4):
        pre = 'transformer.resblocks.%d.' % l
        out += [(pre + 'attn', '.in_proj_weight', '.in_proj_bias'), (pre + 'attn.out_proj', '.weight', '.bias'),
                (pre + 'ln_1', '.weight', '.bias'), (pre + 'mlp.c_fc', '.weight', '.bias'),
                (pre + 'mlp.c_proj', '.weight', '.bias'), (pre + 'ln_2', '.weight', '.bias')]
    out.append(('ln_post', '.weight', '.bias'))
    return out


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleModel(object):
    """the denoiser + schedule of one GaussianDiffusion (reference networks/ddpm.py:168-228)"""

    def __init__(self, weights, dims, hidden_dim, n_types, timesteps=1000, normalize=True,
                 energy_wrapper=False, ebm_per_steps=1, samples_per_step=10, f64=False, model='Diffusion-CCSP'):
        self.L = lib(f64)
        grasp = len(dims) == 3
        self.dims, self.H, self.P, self.T, self.C = dims, hidden_dim, dims[-1][0], timesteps, n_types
        d = Desc(hidden_dim=hidden_dim, pose_dim=dims[-1][0], pose_begin=dims[-1][1], geom_dim=dims[0][0],
                 grasp_dim=dims[1][0] if grasp else 0, grasp_begin=dims[1][1] if grasp else 0,
                 n_types=n_types, timesteps=timesteps, normalize=int(bool(normalize)),
                 energy_wrapper=int(bool(energy_wrapper)), ebm_per_steps=ebm_per_steps,
                 model_kind=int(model == 'StructDiffusion'))
        self.energy_wrapper = bool(energy_wrapper)
        self._keep = []
        ptrs = []
        for name, ws, bs in param_order(n_types, grasp, model == 'StructDiffusion'):
            for suffix in (ws, bs):
                a = _f32(weights[name + suffix])
                self._keep.append(a)
                ptrs.append(a.ctypes.data)
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        h = C.c_void_p()
        self._check(self.L.ccspo_model_create(C.byref(d), arr, C.byref(h)))
        self.h = h
        self.set_schedule(samples_per_step=samples_per_step)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError('oracle: ' + self.L.ccspo_last_error().decode())

    def set_schedule(self, betas=None, step_sizes=None, samples_per_step=10):
        b = None if betas is None else np.ascontiguousarray(betas, dtype=np.float64)
        s = None if step_sizes is None else _f32(step_sizes)
        if np.isscalar(samples_per_step):
            sp, default = None, int(samples_per_step)
        else:
            sp, default = np.ascontiguousarray(samples_per_step, dtype=np.int32), 0
        self._check(self.L.ccspo_schedule_set(self.h, None if b is None else _ptr(b), None if s is None else _ptr(s),
                                              None if sp is None else _ptr(sp), default))

    def set_energy_hook(self, fn):
        """MALA global-batch mode: fn(pair: float64 numpy view of length 2) sums the shard's {E(x), E(x_hat)} over all
        shards in place (e.g. a torch.distributed all_reduce); None removes it"""
        HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double))
        self.L.ccspo_model_set_energy_hook.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        if fn is None:
            self._hook = None
            self._check(self.L.ccspo_model_set_energy_hook(self.h, None, None))
            return

        def trampoline(ctx, pair):
            try:
                fn(np.ctypeslib.as_array(pair, shape=(2,)))
                return 0
            except Exception:           # noqa: never let an exception cross the C boundary
                return 1
        self._hook = HOOK(trampoline)
        self._check(self.L.ccspo_model_set_energy_hook(self.h, C.cast(self._hook, C.c_void_p), None))

    def schedule(self):
        out = {}
        for i, k in enumerate(SCHEDULE_KEYS):
            a = np.empty(self.T, dtype=np.float32)
            self._check(self.L.ccspo_schedule_get(self.h, i, _ptr(a)))
            out[k] = a
        return out

    def time_embedding(self, t):
        a = np.empty(self.H, dtype=np.float32)
        self._check(self.L.ccspo_time_embedding(self.h, int(t), _ptr(a)))
        return a

    def graph(self, batch):
        return OracleGraph(self, batch)

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                self.L.ccspo_model_destroy(self.h)
        except Exception:
            pass


def _np(a):
    if hasattr(a, 'detach'):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


class OracleGraph(object):
    def __init__(self, model, batch):
        self.m = model
        L = model.L
        self.x = _f32(_np(batch.x))
        self.ei = np.ascontiguousarray(_np(batch.edge_index), dtype=np.int64).reshape(2, -1)
        self.ea = _f32(_np(batch.edge_attr))
        self.mask = np.ascontiguousarray(_np(batch.mask), dtype=np.int8)
        self.N, self.F = self.x.shape
        self.E = self.ei.shape[1]
        h = C.c_void_p()
        model._check(L.ccspo_graph_create(model.h, self.N, self.E, self.F, _ptr(self.x), _ptr(self.ei),
                                          _ptr(self.ea), _ptr(self.mask), C.byref(h)))
        self.h = h
        if getattr(batch, 'batch', None) is not None:
            self.seq = np.ascontiguousarray(_np(batch.batch), dtype=np.int64)
            sh = getattr(batch, 'shuffled', None)
            self.shuf = None if sh is None else np.ascontiguousarray(_np(sh), dtype=np.int64)
            model._check(L.ccspo_graph_set_sequences(h, _ptr(self.seq), None if self.shuf is None else _ptr(self.shuf)))

    def denoise(self, poses, t):
        p = _f32(poses)
        out = np.empty_like(p)
        self.m._check(self.m.L.ccspo_denoise(self.m.h, self.h, _ptr(p), int(t), _ptr(out)))
        return out

    def energy_grad(self, poses, t):
        p = _f32(poses)
        g = np.empty_like(p)
        e = np.zeros(1, dtype=np.float32)
        self.m._check(self.m.L.ccspo_energy_grad(self.m.h, self.h, _ptr(p), int(t), _ptr(g), _ptr(e)))
        return g, float(e[0])

    def energy_grad_split(self, poses_enc, poses_tgt, enc_cols, t):
        """one domain of a composed model: the encoder sees poses_enc (first enc_cols columns variable), the energy compares with poses_tgt"""
        pe, pt = _f32(poses_enc), _f32(poses_tgt)
        g = np.empty_like(pe)
        e = np.zeros(1, dtype=np.float32)
        self.m._check(self.m.L.ccspo_energy_grad_split(self.m.h, self.h, _ptr(pe), _ptr(pt), int(enc_cols), int(t), _ptr(g), _ptr(e)))
        return g, float(e[0])

    def edge_outputs(self, poses, t):
        p = _f32(poses)
        out = np.empty((self.E, 2, self.m.P), dtype=np.float32)
        self.m._check(self.m.L.ccspo_edge_outputs(self.m.h, self.h, _ptr(p), int(t), _ptr(out)))
        return out

    def chain(self, sampler, seed=None, normal=None, uniform=None, x=None, t_first=None, t_last=0,
              history=False, row_offset=0, call_base=0, ucall_base=0, accept=False):
        """run timesteps t_first..t_last; x=None draws the initial state (full chain by default)"""
        m = self.m
        nz = Noise()
        keep = []
        if normal is not None:
            nz.mode = 1
            a = _f32(normal)
            keep.append(a)
            nz.normal = a.ctypes.data
            nz.n_normal = a.shape[0]
            if uniform is not None:
                b = _f32(uniform)
                keep.append(b)
                nz.uniform = b.ctypes.data
                nz.n_uniform = b.shape[0]
            nz.call_base, nz.ucall_base = call_base, ucall_base
        else:
            nz.mode, nz.seed, nz.row_offset = 0, int(seed), int(row_offset)
        init = x is None
        xb = np.zeros((self.N, m.P), dtype=np.float32) if init else _f32(x).copy()
        tf = m.T - 1 if t_first is None else int(t_first)
        hist = np.full((m.T + 1, self.N, m.P), np.nan, dtype=np.float32) if history else None
        acc = np.zeros(m.T, dtype=np.float32) if accept else None
        m._check(m.L.ccspo_chain_run(m.h, self.h, SAMPLERS[sampler], C.byref(nz), _ptr(xb), int(init), tf,
                                     int(t_last), None if hist is None else _ptr(hist),
                                     None if acc is None else _ptr(acc)))
        out = [xb]
        if history:
            out.append(hist)
        if accept:
            out.append(acc)
        return out[0] if len(out) == 1 else tuple(out)

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                self.m.L.ccspo_graph_destroy(self.h)
        except Exception:
            pass


def load_weights(path):
    """tests/golden/weights_*.npz -> {reference state_dict key: fp32 array} (int8 rows dequantised)"""
    z = np.load(path)
    out = {}
    for k in z.files:
        if k.endswith('::q8'):
            base = k[:-4]
            out[base] = (z[k].astype(np.float32) * z[base + '::scale'][:, None].astype(np.float32)).astype(np.float32)
        elif k.endswith('::scale'):
            continue
        else:
            out[k] = z[k].astype(np.float32)
    return out
