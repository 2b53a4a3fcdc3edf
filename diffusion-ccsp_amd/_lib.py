"""ctypes binding of libccsp_hip.so (include/ccsp.h) -- the only way the Python host code reaches
the HIP kernels.  There is NO CPU fallback: if the library cannot be loaded (or built) every
entry point raises, loudly."""
import ctypes as C
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
# CCSP_EXPERIMENTS=1 (read once, at import): bind the EXPERIMENTS build instead -- the same library compiled with -DCCSP_EXPERIMENTS, which adds the
# variants that lost their A/Bs (csrc/ccsp_fused.h, row GEMM MODEs 1/2/3/5/7, the node update in the edge kernel's tail, hipGraph replay, CU masks,
# ...) and the switches that select them.  The product build has none of that code (`pytest -m gpu_experiments` covers it, outside the default run).
EXPERIMENTS = os.environ.get('CCSP_EXPERIMENTS', '0') not in ('', '0')
SO = os.path.join(CSRC, 'libccsp_hip_exp.so' if EXPERIMENTS else 'libccsp_hip.so')
SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith(('.h', '.hip'))) + [os.path.join('..', '..', 'include', 'ccsp.h')]     # the one translation unit ccsp_hip.hip and the fragments / kernel headers it includes
ABI_MAJOR = 1         # CCSP_VERSION_MAJOR of include/ccsp.h this binding was written against: lib() refuses a library of another major version

K_COUNT = 12          # CCSP_K_COUNT of include/ccsp.h (a 1.0 library has 11: kernel_stats asks it for its own count)
SAMPLERS = {False: 0, None: 0, 'NONE': 0, 'ULA': 1, 'ULA+': 2, 'MALA': 3, 'HMC': 4}
SCHEDULE_KEYS = ['betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_recip_alphas_cumprod',
                 'sqrt_recipm1_alphas_cumprod', 'posterior_log_variance_clipped', 'posterior_mean_coef1',
                 'posterior_mean_coef2', '_sqrt_recipm1_alphas_cumprod_custom', 'step_sizes',
                 'posterior_variance', 'sqrt_alphas_cumprod', 'sqrt_one_minus_alphas_cumprod',
                 'log_one_minus_alphas_cumprod']
# the twelve buffers GaussianDiffusion registers, in registration order (networks/ddpm.py:200-228): a checkpoint's
# 'model' dict holds exactly these next to the denoise_fn.* weights, and Trainer.load is a strict load_state_dict
REGISTERED_BUFFERS = ['betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod',
                      'sqrt_one_minus_alphas_cumprod', 'log_one_minus_alphas_cumprod', 'sqrt_recip_alphas_cumprod',
                      'sqrt_recipm1_alphas_cumprod', 'posterior_variance', 'posterior_log_variance_clipped',
                      'posterior_mean_coef1', 'posterior_mean_coef2']


class CcspError(RuntimeError):
    pass


ENERGY_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)      # (ctx, device float[2], stream) -> 0 / error


class ModelDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'hidden_dim', 'pose_dim', 'pose_begin', 'geom_dim', 'grasp_dim', 'grasp_begin', 'n_types',
        'timesteps', 'normalize', 'energy_wrapper', 'ebm_per_steps', 'model_kind')]


class Compose(C.Structure):
    """ccsp_compose of include/ccsp.h"""
    _fields_ = [('zero_col', C.c_int32), ('weight_first', C.c_float), ('weight_second', C.c_float), ('normalize', C.c_int32)]


class Noise(C.Structure):
    _fields_ = [('mode', C.c_int32), ('_pad', C.c_int32), ('seed', C.c_uint64), ('row_offset', C.c_uint64),
                ('normal', C.c_void_p), ('n_normal', C.c_uint64), ('uniform', C.c_void_p),
                ('n_uniform', C.c_uint64), ('call_base', C.c_uint64), ('ucall_base', C.c_uint64)]


def _stale(so=None):
    so = so or SO
    if not os.path.isfile(so):
        return True
    t = os.path.getmtime(so)
    return any(os.path.isfile(os.path.join(CSRC, s)) and os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


def build(force=False, verbose=False, experiments=None):
    """hipcc --offload-arch=gfx950 the library in-tree (cross-compiles without a GPU).  experiments (default: this process's CCSP_EXPERIMENTS):
    compile with -DCCSP_EXPERIMENTS into libccsp_hip_exp.so.

    The build REFUSES a binary in which a guarded kernel spills to scratch or in which the compiler placed an instruction on a register whose
    inline-asm load is still in flight (_asmlint.py): both break the hand-counted s_waitcnt of the prefetch kernels.  A different hipcc / ROCm
    point release may trip either check on code that is in fact fine (renamed kernels, a lint false positive).  Escape hatch:
    CCSP_BUILD_SKIP_LINT=1 turns both refusals into warnings -- run `pytest -m gpu` on such a build before trusting it: the parity suite, the
    bitwise-repeatability tests and tools/soak.py are then the only guard."""
    import sys
    import tempfile
    from . import _asmlint
    if experiments is None:
        experiments = EXPERIMENTS
    so = os.path.join(CSRC, 'libccsp_hip_exp.so' if experiments else 'libccsp_hip.so')
    if not force and not _stale(so):
        return so
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    skip_lint = os.environ.get('CCSP_BUILD_SKIP_LINT', '0') not in ('', '0')
    tmp = '%s.%d.tmp' % (so, os.getpid())         # per process: the ranks of a first multi-rank run may all build at once

    def refuse(msg):
        if skip_lint:
            print('diffusion_ccsp_amd build WARNING (CCSP_BUILD_SKIP_LINT=1, not refused): ' + msg, file=sys.stderr)
            return
        raise CcspError(msg + '\n(CCSP_BUILD_SKIP_LINT=1 builds anyway, with this as a warning; then run pytest -m gpu)')
    try:
        with tempfile.TemporaryDirectory(prefix='ccsp_build_') as work:      # -save-temps leaves the device assembly there: the lint reads it
            # --offload-compress: the gfx950 code object travels zstd / zlib-compressed inside the .so (2.0 MB -> 0.5 MB; the HIP runtime unpacks it when
            # the module is loaded, once per process)
            cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-pthread', '-save-temps', '--offload-compress',
                   '-Rpass-analysis=kernel-resource-usage'] + (['-DCCSP_EXPERIMENTS'] if experiments else []) + \
                  ['-o', tmp, os.path.join(CSRC, 'ccsp_hip.hip'), '-ldl']
            if verbose:
                print(' '.join(cmd))
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, cwd=work)
            if r.returncode != 0:
                raise CcspError('hipcc failed:\n' + r.stderr[-4000:])
            bad = check_no_scratch(r.stderr)
            if bad:
                refuse('register spills in kernels whose prefetch loads are inline asm with hand-counted s_waitcnt (a spill or reload next to them '
                       'touches registers that are still in flight, and adds vector-memory operations to the counts): %s' % bad)
            asm = [f for f in os.listdir(work) if f.endswith('gfx950.s')]
            if not asm:
                refuse('hipcc -save-temps left no device assembly in %s: the in-flight register lint cannot run' % work)
            else:
                checked, found = _asmlint.lint_text(open(os.path.join(work, asm[0])).read())
                if found or checked == 0:
                    lines = ['%s line %d: %s (registers %s)' % (k[:80], ln, v[0][:60], v[1][:6]) for k, f in found.items() for ln, v in sorted(f.items())[:3]]
                    refuse('the compiler placed instructions on registers whose inline-asm loads are still in flight (in front of the hand-counted '
                           's_waitcnt; %d kernels checked): \n  %s\n(diffusion-ccsp_amd/_asmlint.py; usual cause: a wait inside a branch, DESIGN 4.7)'
                           % (checked, '\n  '.join(lines[:12]) or 'no guarded kernel found in the assembly'))
        os.replace(tmp, so)
    finally:
        if os.path.isfile(tmp):
            os.remove(tmp)
    return so


# kernels that request operands by inline-asm loads the compiler cannot see and wait for them with hand-counted s_waitcnt
# (csrc/ccsp_f16x2.h h2_ld16 / h2_ld_wait, csrc/ccsp_fused.h fz_ld_frag): a scratch spill there adds vector-memory operations to the
# counts and can store a register whose load is still in flight, so the build refuses a compiler that spills in them
GUARDED_KERNELS = ('k_rowgemm_h2', 'k_edge_h2', 'k_edge_bwd_h2', 'k_node_direct', 'k_node_energy_h2', 'k_eval_fused', 'k_sd_gemm_h2')


def check_no_scratch(remarks):
    """parse hipcc -Rpass-analysis=kernel-resource-usage output; returns {kernel: scratch bytes per lane} of guarded kernels that spill"""
    import re
    bad = {}
    name = None
    for line in remarks.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            name = m.group(1)
            continue
        m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', line)
        if m and name and int(m.group(1)) > 0 and any(k in name for k in GUARDED_KERNELS):
            bad[name] = int(m.group(1))
    return bad


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(SO):
        try:
            build()
        except Exception as e:  # noqa
            raise CcspError('libccsp_hip.so is missing and could not be built (%s); the HIP extension is '
                            'required -- there is no CPU fallback' % (e,))
    try:
        L = C.CDLL(SO)
    except OSError as e:
        raise CcspError('cannot load %s: %s (the HIP extension is required)' % (SO, e))
    vp, i32, u64p = C.c_void_p, C.c_int32, C.POINTER(C.c_uint64)
    L.ccsp_last_error.restype = C.c_char_p
    L.ccsp_version.restype = C.c_int32
    ver = int(L.ccsp_version())
    if ver // 1000 != ABI_MAJOR:         # a stale .so (or a newer header): signatures differ, every call below would pass arguments in the wrong places
        raise CcspError('%s reports ABI version %d.%d, this binding is written against major version %d: rebuild it (diffusion_ccsp_amd.build(force=True))'
                        % (SO, ver // 1000, ver % 1000, ABI_MAJOR))
    L.ccsp_device_info.argtypes = [C.c_char_p, i32, C.POINTER(i32), u64p]
    L.ccsp_model_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(vp), vp, C.POINTER(vp)]
    L.ccsp_model_destroy.argtypes = [vp]
    L.ccsp_model_destroy.restype = None
    L.ccsp_schedule_set.argtypes = [vp, i32, vp, vp, vp, i32]
    L.ccsp_schedule_get.argtypes = [vp, i32, vp]
    L.ccsp_time_embedding.argtypes = [vp, i32, vp, vp]
    L.ccsp_encode.argtypes = [vp, i32, i32, vp, vp, vp]
    L.ccsp_time_mlp.argtypes = [vp, i32, vp, vp, vp]
    L.ccsp_process_constraint.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp]
    L.ccsp_graph_create.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, C.POINTER(vp)]
    L.ccsp_graph_destroy.argtypes = [vp]
    L.ccsp_graph_set_sequences.argtypes = [vp, vp, vp, vp]
    L.ccsp_graph_destroy.restype = None
    L.ccsp_denoise.argtypes = [vp, vp, vp, i32, vp, vp]
    L.ccsp_energy_grad.argtypes = [vp, vp, vp, i32, vp, vp, vp]
    L.ccsp_edge_outputs.argtypes = [vp, vp, vp, i32, vp, vp]
    L.ccsp_chain_run.argtypes = [vp, vp, i32, C.POINTER(Noise), vp, i32, i32, i32, vp, vp, vp]
    L.ccsp_profile_enable.argtypes = [vp, i32]
    L.ccsp_chain_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ccsp_model_set_energy_hook.argtypes = [vp, vp, vp]
    L.ccsp_model_set_energy_allreduce.argtypes = [vp, vp]
    L.ccsp_rccl_unique_id.argtypes = [vp]
    L.ccsp_rccl_comm_create.argtypes = [i32, i32, vp, C.POINTER(vp)]
    L.ccsp_rccl_comm_destroy.argtypes = [vp]
    L.ccsp_rccl_comm_count.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.ccsp_rccl_allreduce_sum_f32.argtypes = [vp, vp, C.c_int64, vp]
    L.ccsp_kernel_stats.argtypes = [vp, i32, C.POINTER(C.c_int64), C.POINTER(C.c_float), C.c_char_p, i32]
    L.ccsp_chain_skipped.argtypes = [vp, C.POINTER(C.c_int64)]
    L.ccsp_chain_margins.argtypes = [vp, vp, C.c_int64]
    if ver % 1000 >= 1:
        L.ccsp_chain_lanes.argtypes = [vp, C.POINTER(i32)]
    L.ccsp_graph_variant.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.ccsp_compose_denoise.argtypes = [vp, vp, vp, vp, C.POINTER(Compose), vp, i32, vp, vp]
    L.ccsp_compose_energy_grad.argtypes = [vp, vp, vp, vp, C.POINTER(Compose), vp, i32, vp, vp, vp]
    L.ccsp_compose_chain_run.argtypes = [vp, vp, vp, vp, C.POINTER(Compose), i32, C.POINTER(Noise), vp, i32, i32, i32, vp, vp, vp]
    L.ccsp_plan_host.argtypes = [i32, i32, i32] + [vp] * 14
    if EXPERIMENTS:
        L.ccsp_plan_fused_host.argtypes = [i32, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp]
    L.ccsp_plan_bwdsum_host.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    if ver % 1000 >= 1:
        L.ccsp_plan_bwdsum_blocks_host.argtypes = [i32, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise CcspError(lib().ccsp_last_error().decode())


def device_info():
    L = lib()
    name = C.create_string_buffer(256)
    cu = C.c_int32()
    mem = C.c_uint64()
    check(L.ccsp_device_info(name, 256, C.byref(cu), C.byref(mem)))
    return dict(name=name.value.decode(), compute_units=cu.value, hbm_bytes=mem.value)


def plan_host(n_nodes, n_types, edge_index, edge_attr):
    """host-only planning tables (no GPU needed); numpy in, dict of numpy out"""
    import numpy as np
    L = lib()
    ei = np.ascontiguousarray(edge_index, dtype=np.int64).reshape(2, -1)
    ea = np.ascontiguousarray(edge_attr, dtype=np.float32)
    E = ei.shape[1]
    names = ['e_orig', 'e_type', 'e_u0', 'e_u1', 'urow_node', 'urow_ts', 'tile_row0', 'tile_nrows', 'tile_ts',
             'node_ptr', 'node_ent']
    sizes = [E, E, E, E, 2 * E, 2 * E, 2 * E + 2 * n_types, 2 * E + 2 * n_types, 2 * E + 2 * n_types,
             n_nodes + 1, 2 * E]
    arrs = [np.full(max(s, 1), -1, dtype=np.int32) for s in sizes]
    counts = np.zeros(3, dtype=np.int32)
    check(L.ccsp_plan_host(n_nodes, E, n_types, ei.ctypes.data, ea.ctypes.data, counts.ctypes.data,
                           *[a.ctypes.data for a in arrs]))
    e_act, rows, tiles = [int(v) for v in counts]
    trims = [e_act, e_act, e_act, e_act, rows, rows, tiles, tiles, tiles, n_nodes + 1, 2 * e_act]
    out = {n: a[:t].copy() for n, a, t in zip(names, arrs, trims)}
    out.update(E_act=e_act, R=rows, n_tiles=tiles)
    return out


def plan_fused_host(n_nodes, n_types, edge_index, edge_attr, rows_per_slot=28, max_edges=112):
    """host-only fused tiles of the one-launch evaluation kernel (include/ccsp.h ccsp_plan_fused_host); numpy in, dict out"""
    import numpy as np
    L = lib()
    ei = np.ascontiguousarray(edge_index, dtype=np.int64).reshape(2, -1)
    ea = np.ascontiguousarray(edge_attr, dtype=np.float32)
    E = ei.shape[1]
    n = C.c_int32()
    tiles = np.zeros((max(E, 1), 4), dtype=np.int32)
    rows = np.zeros((max(E, 1), 128), dtype=np.int32)
    e_lu = np.zeros(max(E, 1), dtype=np.uint16)
    check(L.ccsp_plan_fused_host(n_nodes, E, n_types, ei.ctypes.data, ea.ctypes.data, rows_per_slot, max_edges, C.byref(n), tiles.ctypes.data,
                                 rows.ctypes.data, e_lu.ctypes.data))
    nt = n.value
    e_act = int(tiles[:nt, 2].sum())
    return dict(n_tiles=nt, tiles=tiles[:nt].copy(), rows=rows[:nt].copy(), e_lu=e_lu[:e_act].copy())


def plan_bwdsum_host(n_nodes, n_types, edge_index, edge_attr, block_edges=64, max_parts=128):
    """host-only partial rows of the energy backward (include/ccsp.h ccsp_plan_bwdsum_host / ccsp_plan_bwdsum_blocks_host: (64, 128) = round 4's
    decoder backward, (32, 64) = round 6's fused decoder kernel); numpy in, dict out"""
    import numpy as np
    L = lib()
    ei = np.ascontiguousarray(edge_index, dtype=np.int64).reshape(2, -1)
    ea = np.ascontiguousarray(edge_attr, dtype=np.float32)
    E = ei.shape[1]
    nb, npart = C.c_int32(), C.c_int32()
    blocks = np.zeros((E // block_edges + 1, 1 + 4 * max_parts), dtype=np.int32)
    prow = np.zeros(max(2 * E, 1), dtype=np.int32)
    nptr = np.zeros(n_nodes + 1, dtype=np.int32)
    nidx = np.zeros(max(2 * E, 1), dtype=np.int32)
    if (block_edges, max_parts) == (64, 128):
        check(L.ccsp_plan_bwdsum_host(n_nodes, E, n_types, ei.ctypes.data, ea.ctypes.data, C.byref(nb), C.byref(npart), blocks.ctypes.data,
                                      prow.ctypes.data, nptr.ctypes.data, nidx.ctypes.data))
    else:
        check(L.ccsp_plan_bwdsum_blocks_host(n_nodes, E, n_types, ei.ctypes.data, ea.ctypes.data, block_edges, max_parts, C.byref(nb), C.byref(npart),
                                             blocks.ctypes.data, prow.ctypes.data, nptr.ctypes.data, nidx.ctypes.data))
    return dict(n_blocks=nb.value, NP=npart.value, blocks=blocks[:nb.value].copy(), prow_urow=prow[:npart.value].copy(), nrow_ptr=nptr,
                nrow_idx=nidx[:npart.value].copy())
