"""Host-side mirror of the reference's ``GaussianDiffusion`` sampling interface
(networks/ddpm.py:168-351) on top of the HIP library: same constructor arguments, the same
schedule buffers, ``p_sample_loop(batch, return_history=...)`` and ``sample(batch, **kw)`` with the
``sample_loop_time`` bookkeeping.  The reverse chain itself -- T x (1 + S) network evaluations
with the fused ancestral / Langevin updates -- is one ``ccsp_chain_run`` call that enqueues every
kernel on the current stream without a host synchronisation.

Noise: the reference consumes torch's global generator.  Here the chain's draws come from the
build-owned counter-based stream (noise.py): ``sample(batch, seed=...)``; if no seed is given one
is drawn from torch's global generator, so ``torch.manual_seed`` keeps runs reproducible.
``sample(batch, noise=(normal[, uniform]))`` injects recorded draws instead (parity runs).
"""
import ctypes as C
import sys
import time

import numpy as np
import torch

from . import _lib
from .denoise_fn import _ptr, _stream_ptr


class GaussianDiffusion(object):
    def __init__(self, denoise_fn, timesteps=100, loss_type='l2', EBM=False, betas=None,
                 samples_per_step=10, step_sizes='2*self.betas'):
        self.denoise_fn = denoise_fn
        self.device = denoise_fn.device
        self.dims = denoise_fn.dims
        self.input_mode = denoise_fn.input_mode
        self.loss_type = loss_type
        self.EBM = EBM
        self.training = False
        if betas is not None:
            betas = betas.detach().cpu().numpy() if isinstance(betas, torch.Tensor) else np.asarray(betas)
            timesteps = int(betas.shape[0])
        self.num_timesteps = int(timesteps)
        self.samples_per_step = samples_per_step
        self._core()._bind(self.num_timesteps)
        self._betas_arg = None if betas is None else np.ascontiguousarray(betas, dtype=np.float64)
        self._step_sizes_expr = step_sizes
        self._apply_schedule()
        self.sample_loop_time = []
        self.last_stats = None
        self.record_margins = False       # debugging aid: see last_margins
        self.last_margins = None

    # the native handle lives on the ConstraintDiffuser (possibly behind the EBM wrapper)
    def _core(self):
        inner = getattr(self.denoise_fn, 'model', None)
        return inner if hasattr(inner, '_handle') else self.denoise_fn

    def _apply_schedule(self):
        L = _lib.lib()
        h = self._core()._handle()
        b = self._betas_arg
        sps = self.samples_per_step
        if torch.is_tensor(sps):
            sps = sps.cpu().numpy()
        T = self.num_timesteps
        if np.isscalar(sps):
            sp_arr, default = None, int(sps)
        else:
            sp_arr, default = np.ascontiguousarray(sps, dtype=np.int32), 0
            if sp_arr.shape != (T,):
                raise ValueError('samples_per_step has shape %s, expected (%d,)' % (sp_arr.shape, T))
        if b is not None and b.shape != (T,):
            raise ValueError('betas has shape %s, expected (%d,)' % (b.shape, T))
        _lib.check(L.ccsp_schedule_set(h, T, None if b is None else b.ctypes.data, None,
                                       None if sp_arr is None else sp_arr.ctypes.data, default))
        self._read_buffers()
        # `step_sizes` is a Python expression of self.betas evaluated in the module (ddpm.py:207)
        self.step_sizes = eval(self._step_sizes_expr) if isinstance(self._step_sizes_expr, str) else self._step_sizes_expr
        ss = torch.as_tensor(self.step_sizes, dtype=torch.float32).cpu().numpy()
        ss = np.ascontiguousarray(np.broadcast_to(ss, (self.num_timesteps,)), dtype=np.float32)
        _lib.check(L.ccsp_schedule_set(h, T, None if b is None else b.ctypes.data, ss.ctypes.data,
                                       None if sp_arr is None else sp_arr.ctypes.data, default))
        self._schedule_owner = self._core()._generation

    def _read_buffers(self):
        L = _lib.lib()
        h = self._core()._handle()
        for i, k in enumerate(_lib.SCHEDULE_KEYS):
            a = np.empty(self.num_timesteps, dtype=np.float32)
            _lib.check(L.ccsp_schedule_get(h, i, a.ctypes.data))
            if k != 'step_sizes':
                setattr(self, k, torch.from_numpy(a).to(self.device))

    def _handle(self):
        core = self._core()
        if core.timesteps != self.num_timesteps:
            # two GaussianDiffusion objects with different `timesteps` on ONE ConstraintDiffuser: the native model (time
            # table, schedule) has one length -- re-bind it to this object's rather than replaying a wrong-length schedule
            core._bind(self.num_timesteps)
        h = core._handle()
        if getattr(self, '_schedule_owner', None) != self._core()._generation:   # weights were reloaded -> new native model
            self._apply_schedule()
            h = self._core()._handle()
        return h

    def eval(self):
        self.training = False
        return self

    def load_state_dict(self, sd, strict=True):
        """a reference checkpoint's 'model' dict (Trainer.load, ddpm.py:503-514): denoiser weights are loaded; the stored
        `betas` must describe the schedule this object was built with (the reference would overwrite its buffers with
        them while keeping num_timesteps, step_sizes and the custom kappa of the constructor -- a silent mix); if they
        differ, the stored betas are adopted through the constructor path so that every derived buffer follows them."""
        if 'betas' in sd:
            stored = sd['betas']
            stored = stored.detach().cpu().numpy() if torch.is_tensor(stored) else np.asarray(stored)
            if stored.shape != (self.num_timesteps,):
                raise ValueError('checkpoint betas have shape %s, this GaussianDiffusion has %d timesteps' % (stored.shape, self.num_timesteps))
            if not np.array_equal(stored.astype(np.float32), self.betas.cpu().numpy()):
                self._betas_arg = np.ascontiguousarray(stored, dtype=np.float64)
                self._apply_schedule()
        present = [k for k in _lib.REGISTERED_BUFFERS if k in sd]
        if strict and present and len(present) != len(_lib.REGISTERED_BUFFERS):
            # (a dict with NO schedule buffer is taken as weights-only -- a convenience the reference does not have)
            raise KeyError('missing schedule buffers in state_dict: %s' % ', '.join(k for k in _lib.REGISTERED_BUFFERS if k not in sd))
        self._core().load_state_dict({k: v for k, v in sd.items() if k.startswith('denoise_fn.')}, strict)
        return self

    def state_dict(self):
        """the key set of the reference module's state_dict: its twelve registered buffers (ddpm.py:200-228) and
        the denoiser weights, so that a checkpoint written here loads strictly in the reference's Trainer.load"""
        self._handle()
        out = {k: getattr(self, k) for k in _lib.REGISTERED_BUFFERS}
        pre = 'denoise_fn.model.' if self._core() is not self.denoise_fn else 'denoise_fn.'
        out.update({pre + k: v for k, v in self._core().state_dict().items()})
        return out

    # ------------------------------------------------------------------------------------
    def _sampler(self):
        if not self.EBM:
            return 'NONE'
        if self.EBM in ('ULA', 'ULA+', 'MALA', 'HMC'):
            return self.EBM
        if 'ULA' in self.EBM:
            return 'ULA'
        raise NotImplementedError('EBM=%r' % (self.EBM,))

    def n_normal_calls(self):
        """randn(N, P) draws of one full chain: the initial state, one per ancestral step, and the sampler's own draws on the
        timesteps where it runs (j % ebm_per_steps == 0, ddpm.py:330)"""
        T = self.num_timesteps
        eps = max(1, int(getattr(self.denoise_fn, 'ebm_per_steps', 1)))
        active = [t for t in range(T) if t % eps == 0]
        kind = self._sampler()
        if kind == 'NONE':
            return 1 + T
        if kind == 'ULA+':
            n = T // 4
            return 1 + T + sum(4 * (min(3, t // n if n else 3) + 1) for t in active)
        if kind == 'HMC':                                # momentum + 4 refreshments (ddpm.py:1090,1096)
            return 1 + T + 5 * len(active)
        sps = self.samples_per_step
        if torch.is_tensor(sps):
            sps = sps.cpu().numpy()
        if np.isscalar(sps):
            return 1 + T + int(sps) * len(active)
        return 1 + T + int(sum(int(np.asarray(sps)[t]) for t in active))

    def _noise_struct(self, seed, noise, row_offset):
        dev = self.device
        nz = _lib.Noise()
        keep = []
        if noise is not None:
            normal = noise[0] if isinstance(noise, (tuple, list)) else noise
            normal = normal.detach().to(dev, torch.float32).contiguous()
            keep.append(normal)
            nz.mode, nz.normal, nz.n_normal = 1, normal.data_ptr(), normal.shape[0]
            if isinstance(noise, (tuple, list)) and len(noise) > 1 and noise[1] is not None:
                uni = noise[1].detach().to(dev, torch.float32).contiguous()
                keep.append(uni)
                nz.uniform, nz.n_uniform = uni.data_ptr(), uni.shape[0]
        else:
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            nz.mode, nz.seed, nz.row_offset = 0, int(seed), int(row_offset)
        return nz, keep

    def _run(self, batch, x, init, t_first, t_last, return_history, seed, noise, row_offset):
        assert not self.training
        L = _lib.lib()
        core = self._core()
        h = self._handle()
        dev = self.device
        T = self.num_timesteps
        nz, keep = self._noise_struct(seed, noise, row_offset)
        if getattr(core, '_second', None) is not None:
            # two composed domains (ConstraintDiffuser.compose): one evaluation per domain and evaluation, ccsp_compose_chain_run
            if self._sampler() in ('MALA', 'HMC') and not core.energy_wrapper:
                raise NotImplementedError('composed domains run EBM=False, ULA and ULA+ (on the denoiser output, or on the energy gradient '
                                          'when the composed model is an energy_wrapper model) and, as energy_wrapper models, MALA and HMC')
            if core.energy_wrapper and tuple(core.composing_weight) != (1, 1):
                raise NotImplementedError('the energy of composed domains is built for composing_weight (1, 1)')
            first, second = core._composed_parts()
            h = self._handle()
            g1, g2 = core._composed_graphs(batch)
            hist = torch.empty((T + 1, g1.N, self.dims[-1][0]), device=dev, dtype=torch.float32) if return_history else None
            c = core._compose_struct()
            acc = torch.zeros(T, device=dev, dtype=torch.float32) if self._sampler() in ('MALA', 'HMC') else None
            marg = self._margin_buffer(g1, t_first, t_last)
            try:
                with torch.cuda.device(dev):
                    _lib.check(L.ccsp_compose_chain_run(h, g1.h, second._h, g2.h, C.byref(c), _lib.SAMPLERS[self._sampler()], C.byref(nz), _ptr(x),
                                                        int(init), int(t_first), int(t_last), None if hist is None else _ptr(hist),
                                                        None if acc is None else _ptr(acc), _stream_ptr(dev)))
            finally:
                self._release_margins(g1, marg)
            self._last_graph = g1
            self._keepalive = keep + [g2]
            self.last_accept_rates = acc
            self.last_margins = marg
            return hist
        g = core._graph(batch)
        hist = torch.empty((T + 1, g.N, self.dims[-1][0]), device=dev, dtype=torch.float32) if return_history else None
        acc = torch.zeros(T, device=dev, dtype=torch.float32) if self._sampler() in ('MALA', 'HMC') else None
        marg = self._margin_buffer(g, t_first, t_last)
        try:
            with torch.cuda.device(dev):
                _lib.check(L.ccsp_chain_run(h, g.h, _lib.SAMPLERS[self._sampler()], C.byref(nz), _ptr(x), int(init), int(t_first),
                                            int(t_last), None if hist is None else _ptr(hist), None if acc is None else _ptr(acc),
                                            _stream_ptr(dev)))
        finally:
            self._release_margins(g, marg)
        self._last_graph = g
        self._keepalive = keep
        self.last_accept_rates = acc          # MetropolisSampler's per-timestep acceptance (ddpm.py:979-996)
        self.last_margins = marg
        return hist

    def _inner_steps(self, t):
        """accept steps of timestep t under MALA / HMC (samples_per_step; HMC: the reference's fixed 4, ddpm.py:311)"""
        if t % max(1, int(getattr(self.denoise_fn, 'ebm_per_steps', 1))) != 0:
            return 0
        if self._sampler() == 'HMC':
            return 4
        sps = self.samples_per_step
        if torch.is_tensor(sps):
            sps = sps.cpu().numpy()
        return int(sps) if np.isscalar(sps) else int(np.asarray(sps)[t])

    def _margin_buffer(self, g, t_first, t_last):
        """record_margins (debugging aid; ccsp_chain_margins): a [accept steps of this call, 2, N] buffer the accept kernels fill with
        [:, 0] = log(acceptance ratio) - log(u) per node row -- positive = accepted, negative = rejected -- and [:, 1] = the sum of the
        absolute values of the terms of that ratio; |margin| <~ 1e-6 scale = a near-tie that fp32 rounding may decide either way (the
        parity tests assert that every difference from the reference's decisions begins at one).  -> self.last_margins after the chain"""
        if not self.record_margins or self._sampler() not in ('MALA', 'HMC'):
            return None
        k = sum(self._inner_steps(t) for t in range(int(t_last), int(t_first) + 1))
        marg = torch.full((max(k, 1), 2, g.N), float('nan'), device=self.device, dtype=torch.float32)
        _lib.check(_lib.lib().ccsp_chain_margins(g.h, _ptr(marg), marg.numel()))
        return marg

    def _release_margins(self, g, marg):
        """uninstall the margin buffer from the (cached) graph on EVERY exit of a chain call: a chain that raised -- an exhausted injected noise
        stream, a HIP error -- would otherwise leave the graph pointing at `marg`, which is freed when the exception unwinds, and the next MALA /
        HMC chain on that graph would write into freed memory.  What was already enqueued may still write into the buffer: wait for it first."""
        if marg is None:
            return
        if sys.exc_info()[0] is not None:
            try:
                torch.cuda.synchronize(self.device)
            except Exception:                    # noqa: BLE001 -- the original error is the one to report
                pass
        rc = _lib.lib().ccsp_chain_margins(g.h, None, 0)
        if sys.exc_info()[0] is None:
            _lib.check(rc)

    def p_sample_loop(self, batch, return_history=False, seed=None, noise=None, row_offset=0, **kwargs):
        """GaussianDiffusion.p_sample_loop (ddpm.py:260-340)"""
        T = self.num_timesteps
        self._handle()                         # (binds the denoiser's native model to this object's schedule first)
        x = torch.empty((batch.x.shape[0], self.dims[-1][0]), device=self.device, dtype=torch.float32)
        hist = self._run(batch, x, 1, T - 1, 0, return_history, seed, noise, row_offset)
        if return_history:
            return x, [hist[i] for i in range(T + 1)]
        return x

    def p_sample_segment(self, batch, x, t_first, t_last, seed=None, noise=None, row_offset=0):
        """timesteps t_first..t_last of the loop starting from the state x [N,P] (a chain split across calls
        reproduces the unsplit chain: noise draws are indexed by their call number)"""
        x = x.detach().to(self.device, torch.float32).contiguous().clone()
        self._run(batch, x, 0, t_first, t_last, False, seed, noise, row_offset)
        return x

    def sample(self, batch, **kwargs):
        """GaussianDiffusion.sample (ddpm.py:342-351): wall-clock timed p_sample_loop"""
        start = time.time()
        outputs = self.p_sample_loop(batch, **kwargs)
        torch.cuda.synchronize(self.device)
        passed = time.time() - start
        self.sample_loop_time.append(passed)
        if len(self.sample_loop_time) > 10:
            self.sample_loop_time.pop(0)
        return outputs

    def chain_stats(self):
        """HIP-event timing of the last chain on its graph (ccsp_chain_stats)"""
        g = getattr(self, '_last_graph', None)
        if g is None:
            raise _lib.CcspError('no chain has run')
        ev, ms, mu, me = C.c_int64(), C.c_float(), C.c_float(), C.c_float()
        _lib.check(_lib.lib().ccsp_chain_stats(g.h, C.byref(ev), C.byref(ms), C.byref(mu), C.byref(me)))
        sk = C.c_int64()
        _lib.check(_lib.lib().ccsp_chain_skipped(g.h, C.byref(sk)))
        return dict(evals=ev.value, ms_total=ms.value, ms_ugemm=mu.value, ms_edge=me.value, evals_skipped=sk.value)

    def chain_lanes(self):
        """concurrent lanes (sub-batches on streams of their own, one enqueueing host thread each) the last chain ran as (ccsp_chain_lanes)"""
        g = getattr(self, '_last_graph', None)
        if g is None:
            raise _lib.CcspError('no chain has run')
        n = C.c_int32()
        try:
            fn = _lib.lib().ccsp_chain_lanes
        except AttributeError:                  # (ABI 1.0 library: tools/mkcommit.sh builds of earlier commits)
            return None
        _lib.check(fn(g.h, C.byref(n)))
        return int(n.value)

    def kernel_stats(self):
        """per-kernel launch counts and mean durations (ms) of the last PROFILED chain: {label: (calls, ms_mean)}"""
        g = getattr(self, '_last_graph', None)
        if g is None:
            raise _lib.CcspError('no chain has run')
        out = {}
        for k in range(_lib.K_COUNT):
            n, ms, name = C.c_int64(), C.c_float(), C.create_string_buffer(64)
            rc = _lib.lib().ccsp_kernel_stats(g.h, k, C.byref(n), C.byref(ms), name, 64)
            if rc != 0 and k >= 11:             # (an ABI 1.0 library -- tools/mkcommit.sh builds of earlier commits -- has eleven selectors)
                continue
            _lib.check(rc)
            if n.value:
                out[name.value.decode()] = (int(n.value), float(ms.value))
        return out

    def kernel_variant(self):
        """(MODE of k_rowgemm_h2, edges per workgroup of the edge kernel) a one-lane launch on the last chain's graph runs"""
        g = getattr(self, '_last_graph', None)
        if g is None:
            raise _lib.CcspError('no chain has run')
        rm, et = C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().ccsp_graph_variant(g.h, C.byref(rm), C.byref(et)))
        return int(rm.value), int(et.value)

    def profile(self, batch, on=True):
        """bracket the evaluation kernels of the next chains on this batch with HIP events"""
        g = self._core()._graph(batch)
        _lib.check(_lib.lib().ccsp_profile_enable(g.h, int(bool(on))))
