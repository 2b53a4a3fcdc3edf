#!/usr/bin/env python
"""bench.py -- samples/sec (and solved samples/sec) of the Diffusion-CCSP reverse-diffusion sampler on MI355X.

Workloads (BASELINE.json configs; hyper-parameters are the reference defaults, train_utils.py:86-89: hidden_dim 256,
samples_per_step 10, step_sizes '2*self.betas'), one per-GPU shard each, synthetic graphs:

  --config c2 (default; c3 = the same shard on every GPU)
        RandomSplitQualitativeWorld, 8 objects, T=1000 ULA, 256 graphs per GPU -- 11 000 network evaluations per chain
  --config c4   TriangularRandomSplitWorld, 12 objects, T=1000 MALA (energy mode), 256 graphs per GPU (1024 over 4 GPUs;
                each shard is its own reference batch: replica semantics)
  --config c5   3D panda-box packing (robot_box), 10 objects, T=1000 ULA, 64 graphs per GPU (512 over 8 GPUs)

A "step" is one whole `GaussianDiffusion.sample(batch)` call: graph upload / planning + the full reverse chain of the
rank's graphs (+ the gather of final poses when N > 1).  Inputs (the collated batch tensors and the weights) are resident
in HBM before the timed region.

    python bench.py --gpus 1 --steps 3 --warmup 1 [--config c2|c4|c5]
    python bench.py --gpus N --steps K --warmup W          N > 1 with WORLD_SIZE unset: re-executes ITSELF as N ranks under
                                                           torch.distributed.run (127.0.0.1, a free port); one rank per GPU over RCCL
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          (the same, launched from outside)
    python bench.py --gpus 2 --backend gloo --dry-run      the N > 1 host path WITHOUT a GPU: launcher, init_process_group, flat weight
                                                           broadcast, shard bookkeeping, gather, per-rank timing, the JSON line
                                                           ("dry_run": true, value null -- no sampling happens: there is no CPU fallback)

    python bench.py --gpus N --scaling strong     C3 proper: 2048 graphs in total, split over the N ranks (default: weak, 256 per GPU)

Rank 0 prints ONE JSON line.  Extra blocks:
  "roofline"      every kernel of the chain timed launch by launch with HIP events on the chain's own stream in a separate
                  profiled pass (ccsp_kernel_stats); the dominant kernel priced twice -- EXECUTED matrix-pipe flops / its mean
                  duration / the dense peak of the pipe that runs them (frac_mfma), and its fabric-side bytes / duration /
                  8 TB/s (frac_bytes) -- and `bound` names the larger.  The bytes are measured in THIS run: two rocprofv3 --pmc
                  passes (FETCH_SIZE, WRITE_SIZE; separate passes, FETCH_SIZE doubled: MI355X_MICROARCH.md, HBM section) over
                  tools/profile_eval.py on the same batch, kernel symbol and variant checked against what the timed pass ran;
                  if rocprofv3 is not available the committed summary under profiles/ is used instead and stamped with its
                  git blob hash (and refused when it was taken on another kernel variant).  The reference formulation's
                  ALGORITHMIC flops over the same time are reported beside it, not as the fraction: the
                  row factorisation executes 3.7x fewer flops than the reference's per-edge products.
  "cpu_baseline"  the cost-faithful PyTorch-CPU port of the reference sampler (oracle/torch_proxy.py) timed on this box's
                  host cores on a bounded sample.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# kernel arguments in device memory (a chain is 33 000 short dependent launches; ROCm >= 7 default): read by the HIP runtime at its first call,
# so it is set HERE, before torch is imported, and inherited by the ranks of a self-launched N > 1 run (profiles/r03_findings.md: 13-27 %)
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # the host driver only supports dmabuf IPC (RCCL across processes)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# dense matrix-pipe peaks, /opt/skills/guides/MI355X_MICROARCH.md (never the 2:1-sparsity figures)
PEAKS = {'f16': 2500.0, 'bf16': 2500.0, 'f32': 157.3}
HIDDEN = 256
T_STEPS = 1000
S_LANGEVIN = 10

CONFIGS = {
    'c2': dict(mode='qualitative', n_types=13, n_objects=8, graphs=256, EBM='ULA', energy=False, batch='qualitative_batch',
               weights=('weights/qualitative_h256_ref30k.npz', 'tests/golden/weights_qualitative_h256.npz'),
               label='C2: RandomSplitQualitativeWorld 8 objects, T=1000 ULA S=10'),
    'c4': dict(mode='diffuse_pairwise', n_types=2, n_objects=12, graphs=256, EBM='MALA', energy=True, batch='triangular_batch',
               weights=('tests/golden/weights_diffuse_pairwise_h256_energy.npz',),
               label='C4: TriangularRandomSplitWorld 12 objects, T=1000 MALA S=10 (energy mode)'),
    'c5': dict(mode='robot_box', n_types=2, n_objects=10, graphs=64, EBM='ULA', energy=False, batch='robot_box_batch',
               weights=('tests/golden/weights_robot_box_h256.npz',),
               label='C5: 3D panda-box packing (robot_box) 10 objects, T=1000 ULA S=10'),
}
WEIGHT_NOTES = {
    'weights/qualitative_h256_ref30k.npz': 'the reference recipe AS WRITTEN for input_mode qualitative (train_utils.py:87,142-156,217-218; ddpm.py:444,519-556: a fixed set of '
                                           '30 000 worlds of 2-5 objects, shuffled epochs, batch 128, Adam 5e-4, no EMA) on one MI355X by tools/train_gpu.py TRAIN_RECIPE=reference, '
                                           'the 30 000-step checkpoint; the thirteen type-MLP matrices stored int8 per output row (round 5: 34 MB -> 10 MB of what travels to the GPU box; '
                                           'the DEQUANTISED values are the weights -- tests/golden/chain_q256_bench_B16 is the REFERENCE sampling with exactly these).  The recipe runs '
                                           '300 000 steps; the solved rate of its checkpoints peaks at 30-40 k steps and falls afterwards because the reference sampler overflows fp32 on '
                                           'more and more graphs (profiles/r03_solved_curve_reference_recipe*.json, tests/golden/chain_q256_ref300k_B16.npz = the REFERENCE sampling with '
                                           'the 300 000-step weights: 16 of 16 graphs non-finite).  tools/fetch_or_train_weights.sh re-trains any of the checkpoints',
    'tests/golden/weights_qualitative_h256.npz': 'parity fixture: 2000 CPU steps of the reference loss (oracle/ref_train.py)',
    'tests/golden/weights_diffuse_pairwise_h256_energy.npz': 'parity fixture: 1200 CPU steps of the reference loss in energy mode (oracle/ref_train.py); the weights '
                                                             'the reference-generated C4 goldens were made with',
    'tests/golden/weights_robot_box_h256.npz': 'parity fixture: 2000 CPU steps of the reference loss (oracle/ref_train.py) on synthetic robot_box graphs',
}


def load_weights(path):
    """npz with reference state_dict key names; type-MLP matrices may be stored int8 + one fp32 scale per output row
    ('::q8' / '::scale'): the DEQUANTISED values are the weights (the reference goldens were generated from exactly these)"""
    z = np.load(path)
    out = {}
    for k in z.files:
        if k.endswith('::q8'):
            out[k[:-4]] = (z[k].astype(np.float32) * z[k[:-4] + '::scale'][:, None]).astype(np.float32)
        elif not k.endswith('::scale'):
            out[k] = z[k].astype(np.float32)
    return out


def weights_storage(path):
    z = np.load(path)
    return 'int8 per-row-scaled type-MLP matrices (dequantised values are the weights), fp32 elsewhere' if any(k.endswith('::q8') for k in z.files) else 'fp32'


def algorithmic_flops(n_nodes, n_edges, H, P, grasp):
    """reference dense formulation, 2 flop per multiply-add, per network evaluation (direct mode;
    energy mode = 2x: forward + input gradient)"""
    kin = (6 if grasp else 5) * H
    f_node = 2 * (P * H // 2 + (H // 2) * H)
    f_edge = 2 * (kin * 2 * H) + 2 * 2 * (H * H // 2 + (H // 2) * P)
    return n_nodes * f_node + n_edges * f_edge, f_node, f_edge


def backward_mode(mma):
    """GEMM scheme of the energy-mode backward kernels (ccsp_hip.hip reads CCSP_ENERGY_BWD at model creation)"""
    if mma == 'f16x2' and os.environ.get('CCSP_ENERGY_BWD', '') == 'bf16x3':
        return 'bf16x3'
    return mma


def encoder_mode(mma):
    """pipe of the pose encoder's second layer in the node kernels (CCSP_ENC=f32 keeps v_mfma_f32_16x16x4_f32)"""
    return 'f16x2' if mma == 'f16x2' and os.environ.get('CCSP_ENC', '') != 'f32' else 'f32'


def kernel_symbol(label, mma, energy=False):
    row = KERNEL_SYMBOLS.get(label, {})
    if label == 'node update + pose encoder':      # direct-mode chains on the f16x2 kernels run the straight-line form
        return 'k_node_direct' if (mma == 'f16x2' and not energy and os.environ.get('CCSP_NODE', '') != 'generic') else 'k_node<256'
    if label == 'node energy backward':
        return 'k_node_energy_h2' if encoder_mode(mma) == 'f16x2' else 'k_node_energy_mfma'
    return row.get(backward_mode(mma) if label in ('edge decoder backward', 'row GEMM (transpose)') else mma, '')


def executed_work(label, N, E, R, H, mma):
    """(fp32-equivalent flops, matrix-pipe products per fp32 product, pipe) of one launch of the kernel behind a
    ccsp_kernel_stats label -- what the kernels EXECUTE after the row factorisation; None for non-matrix kernels"""
    prod = {'f16x2': (3, 'f16'), 'bf16x3': (6, 'bf16'), 'f32': (1, 'f32')}[mma]
    bwd = {'f16x2': (3, 'f16'), 'bf16x3': (6, 'bf16'), 'f32': (1, 'f32')}[backward_mode(mma)]
    enc = (3, 'f16') if encoder_mode(mma) == 'f16x2' else (1, 'f32')       # the pose encoder's second layer (node kernels)
    table = {
        'row GEMM (forward)': (2.0 * R * (2 * H) * H,) + prod,
        'edge decoder (forward)': (2.0 * (2 * E) * (H // 2) * H,) + prod,
        'node update + pose encoder': (2.0 * N * H * (H // 2),) + enc,
        # energy-mode backward kernels: same scheme as the forward ones unless CCSP_ENERGY_BWD=bf16x3
        'edge decoder backward': (2.0 * (2 * E) * H * (H // 2),) + bwd,
        'edge decoder forward + backward': (2.0 * 2.0 * (2 * E) * H * (H // 2),) + prod,     # (round 6: one kernel, both GEMMs)
        'row GEMM (transpose)': (2.0 * R * H * (2 * H),) + bwd,
        'node energy backward': (2.0 * 2 * N * H * (H // 2),) + enc,
    }
    return table.get(label)


KERNEL_SYMBOLS = {      # label -> kernel symbol prefix by GEMM mode (for the PMC traffic lookup and the report)
    'row GEMM (forward)': {'f16x2': 'k_rowgemm_h2<256, 512', 'bf16x3': 'k_rowgemm_bf2<256, 512', 'f32': 'k_rowgemm<256, 512>'},
    'edge decoder (forward)': {'f16x2': 'k_edge_h2', 'bf16x3': 'k_edge_bf2', 'f32': 'k_edge<256'},     # (k_edge_h2s on small batches: same prefix)
    'node update + pose encoder': {m: 'k_node<256' for m in ('f16x2', 'bf16x3', 'f32')},
    'edge decoder backward': {'f16x2': 'k_edge_bwd_h2', 'bf16x3': 'k_edge_bwd_bf', 'f32': 'k_edge_bwd<256>'},
    'edge decoder forward + backward': {'f16x2': 'k_edge_fb_h2'},
    'row GEMM (transpose)': {'f16x2': 'k_rowgemm_h2<512, 256', 'bf16x3': 'k_rowgemm_bf2<512, 256>', 'f32': 'k_rowgemm<512, 256>'},
    'node energy backward': {m: 'k_node_energy' for m in ('f16x2', 'bf16x3', 'f32')},     # (_h2 / _h2_update / _mfma: same prefix)
    'row sum of g_z': {m: 'k_rowsum' for m in ('f16x2', 'bf16x3', 'f32')},
    'energy sum': {m: 'k_energy_sum' for m in ('f16x2', 'bf16x3', 'f32')},
}


def _git_blob_hash(path):
    import hashlib
    data = open(path, 'rb').read()
    return hashlib.sha1(b'blob %d\0' % len(data) + data).hexdigest()


def pmc_traffic_file(config, symbols):
    """fabric-side bytes per launch of each kernel whose name starts with one of `symbols`, from the COMMITTED rocprofv3 --pmc
    summary of this round (profiles/r05_pmc_<config>.txt, else round 4's; made by tools/pmc_run.sh).  -> ({symbol: {kernel name, bytes}}, stamp)"""
    path = next((q for q in (os.path.join(ROOT, 'profiles', '%s_pmc_%s.txt' % (r, config)) for r in ('r06', 'r05', 'r04')) if os.path.isfile(q)), None)
    if path is None:
        return {}, None
    cur, got = None, {}
    for line in open(path):
        if not line.startswith(' '):
            cur = line.strip()
            continue
        f = line.split()
        if f and f[0] in ('FETCH_SIZE', 'WRITE_SIZE') and cur is not None:
            got.setdefault(cur, {})[f[0]] = float(f[2])
    out = {}
    for sym in symbols:
        for name, d in got.items():
            if name.startswith(sym) and len(d) == 2:
                out[sym] = {'kernel_name': name, 'bytes': d['FETCH_SIZE'] * 1024.0 * 2.0 + d['WRITE_SIZE'] * 1024.0}
    return out, {'source': os.path.relpath(path, ROOT), 'git_blob': _git_blob_hash(path)}


def pmc_traffic_live(config, graphs, symbols, timeout=240):
    """the same figures measured NOW: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two passes: they do not fit one) over
    tools/profile_eval.py -- single evaluations of the same per-GPU batch in a child process -- with nothing but --kernel-trace next
    to the counters.  Units: KiB per dispatch; FETCH_SIZE doubled (16-byte-per-lane reads are tallied at half their bytes on
    gfx950, MI355X_MICROARCH.md HBM section).  Infinity-Cache hits are included: an upper bound on HBM bytes.
    -> ({symbol: {kernel name, bytes}}, stamp) or ({}, reason)"""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.isfile('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return {}, {'source': 'none', 'reason': 'rocprofv3 not found'}
    vals = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='ccsp_pmc_', dir='/tmp')
        try:
            cmd = [exe, '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', d, '--', sys.executable,
                   os.path.join(ROOT, 'tools', 'profile_eval.py'), '6', str(graphs), config]
            r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
            files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
            if r.returncode != 0 or not files:
                return {}, {'source': 'none', 'reason': 'rocprofv3 --pmc %s failed (rc %d)' % (counter, r.returncode)}
            acc = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row['Counter_Name'] == counter:
                        name = row['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
                        acc.setdefault(name, []).append(float(row['Counter_Value']))
            vals[counter] = {k: sum(v) / len(v) for k, v in acc.items()}
        except Exception as e:           # noqa: a profiler problem must not take the benchmark down
            return {}, {'source': 'none', 'reason': 'rocprofv3 --pmc %s: %s' % (counter, e)}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {}
    for sym in symbols:
        for name in vals['FETCH_SIZE']:
            if name.startswith(sym) and name in vals['WRITE_SIZE']:
                out[sym] = {'kernel_name': name, 'bytes': vals['FETCH_SIZE'][name] * 1024.0 * 2.0 + vals['WRITE_SIZE'][name] * 1024.0}
    return out, {'source': 'live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes) over tools/profile_eval.py 6 %d %s in this run' % (graphs, config)}


def rocprof_kernel_stats_live(config, graphs, symbols, timeout=240):
    """rocprofv3 --kernel-trace --stats (no counters) over tools/profile_eval.py on the same batch, in THIS run: the profiler's own AverageNs of
    each kernel, to stand next to the HIP-event us_mean of the timed pass (which spans launch to next mark).  One-lane launches of a few
    timesteps of the chain.  -> ({symbol: {kernel name, calls, avg_ns, min_ns, max_ns}}, stamp) or ({}, reason)"""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.isfile('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return {}, {'source': 'none', 'reason': 'rocprofv3 not found'}
    d = tempfile.mkdtemp(prefix='ccsp_stats_', dir='/tmp')
    try:
        cmd = [exe, '--kernel-trace', '--stats', '--output-format', 'csv', '-d', d, '--', sys.executable,
               os.path.join(ROOT, 'tools', 'profile_eval.py'), '220', str(graphs), config]
        r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
        files = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)
        if r.returncode != 0 or not files:
            return {}, {'source': 'none', 'reason': 'rocprofv3 --kernel-trace --stats failed (rc %d)' % r.returncode}
        out = {}
        for row in csv.DictReader(open(files[0])):
            name = row['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            for sym in symbols:
                if name.startswith(sym) and (sym not in out or int(row['Calls']) > out[sym]['calls']):
                    out[sym] = {'kernel_name': name, 'calls': int(row['Calls']), 'avg_ns': float(row['AverageNs']),
                                'min_ns': float(row['MinNs']), 'max_ns': float(row['MaxNs'])}
        return out, {'source': 'live: rocprofv3 --kernel-trace --stats over tools/profile_eval.py 220 %d %s in this run (one lane)' % (graphs, config)}
    except Exception as e:           # noqa: a profiler problem must not take the benchmark down
        return {}, {'source': 'none', 'reason': 'rocprofv3 --kernel-trace --stats: %s' % e}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def algorithmic_bytes(n_nodes, n_types_present, H, P, grasp):
    """bytes one evaluation must touch -- the weights of the present types + encoders / decoder + state"""
    kin = (6 if grasp else 5) * H
    per_type = (2 * H * kin + 2 * H) * 4
    shared = ((H // 2) * 8 + H * (H // 2) + (H // 2) * P + H * (H // 2) + (H // 2) * H + P * (H // 2) + 3 * H) * 4
    return n_types_present * per_type + shared + 3 * n_nodes * P * 4


def working_set_bytes(n_nodes, E_act, R, H, P, types_present, energy):
    """bytes one evaluation reads or writes at least once (f16x2 kernels): the chain-constant row terms `base`, the pre-activation
    rows U and their maxima, the pose-embedding planes, the edge outputs, the fp16 planes of the present types' pose slices and of the
    decoder; energy mode adds Q, GZ, the row sums and their planes, the transposed weights"""
    ws = 2 * R * 2 * H * 4 + R * 8 * 4 + 2 * n_nodes * H * 2 + 2 * E_act * P * 4 + types_present * 2 * 2 * 2 * H * H * 2 + 2 * (H // 2) * H * 2
    if energy:
        ws += 2 * E_act * (H // 2) * 4 + E_act * 2 * H * 4 + R * 2 * H * 4 + 2 * R * 2 * H * 2 + R * H * 4 + types_present * 2 * 2 * 2 * H * H * 2
    return ws


def throughput_regime(gd, cfg, worlds, dev, args, graphs=1024):
    """the same chain on a batch whose `base` + U (2 x 160 MB) exceed the Infinity Cache: there the fabric-side bytes ARE HBM bytes.
    One profiled chain (kernel durations) + two live --pmc passes on the same batch size -> us per evaluation, HBM-side TB/s of
    the evaluation and of the dominant kernel, its matrix-pipe fraction"""
    import gc
    batch_np = getattr(worlds, cfg['batch'])(graphs, cfg['n_objects'], seed=5)
    b = batch_np.to_torch(dev)
    gd.profile(b, True)
    gd.sample(b, seed=78)
    ks = gd.kernel_stats()
    st = gd.chain_stats()
    gd.profile(b, False)
    del b
    gc.collect()
    from diffusion_ccsp_amd import _lib
    plan = _lib.plan_host(batch_np.x.shape[0], cfg['n_types'], batch_np.edge_index, batch_np.edge_attr)
    R, E_act, N = plan['R'], plan['E_act'], batch_np.x.shape[0]
    P = worlds.MODE_DIMS[cfg['mode']][-1][0]
    out = {'graphs': graphs, 'nodes': N, 'edges': E_act, 'u_rows': R,
           'working_set_bytes': working_set_bytes(N, E_act, R, HIDDEN, P, cfg['n_types'], cfg['energy'])}
    timed = sum(ms * calls for calls, ms in ks.values())
    n_eval = ks.get('row GEMM (forward)', (0, 0.0))[0]
    out['us_per_evaluation'] = 1e3 * timed / n_eval if n_eval else None
    out['chain_ms_event'] = st['ms_total']
    out['kernels'] = {label: {'us_mean': 1e3 * ms, 'calls_timed': calls} for label, (calls, ms) in ks.items()}
    row = ks.get('row GEMM (forward)')
    if row:
        w = executed_work('row GEMM (forward)', N, E_act, R, HIDDEN, 'f16x2')
        out['row_gemm_frac_mfma'] = w[1] * w[0] / (row[1] * 1e-3) / 1e12 / PEAKS[w[2]]
    if not args.no_live_pmc:
        syms = ['k_rowgemm_h2<256, 512', 'k_edge_h2', 'k_node_direct']
        traffic, stamp = pmc_traffic_live('c2', graphs, syms)
        if traffic:
            tot = sum(t['bytes'] for t in traffic.values())
            out['hbm_side_bytes_per_evaluation'] = tot
            out['traffic_source'] = stamp
            if out['us_per_evaluation']:
                out['hbm_side_tb_per_s'] = tot / (out['us_per_evaluation'] * 1e-6) / 1e12
                out['frac_of_8_tb_per_s'] = out['hbm_side_tb_per_s'] / 8.0
            tr = traffic.get('k_rowgemm_h2<256, 512')
            if tr and row:
                out['row_gemm_bytes_per_launch'] = tr['bytes']
                out['row_gemm_tb_per_s'] = tr['bytes'] / (row[1] * 1e-3) / 1e12
        else:
            out['traffic_source'] = stamp
    out['note'] = ('%d graphs in one lane: base + U = %.0f MB do not fit the 256 MiB Infinity Cache, so 2 x FETCH_SIZE + WRITE_SIZE is HBM traffic here; '
                   'the configuration\'s own 256 graphs per GPU sit below that (roofline.bound = fabric / mfma)' % (graphs, 2 * R * 2 * HIDDEN * 4 / 1e6))
    return out


def sd_main(args):
    """--config sd: the StructDiffusion transformer baseline (denoise_fn.py:391-451, transformer.py:43-82) on 256 graphs x 7
    objects (8-token sequences), T=1000 ULA S=10, hidden_dim 256 (transformer width 512, 4 blocks, 2 heads).  Not a BASELINE.json
    configuration: the place north_star attaches "MFMA for the dense transformer linear layers" to.  Random-init weights (nn.Linear
    default init; the reference ships no checkpoint and timing is value-independent).  Graphs are independent: N ranks = N replicas of
    the shard, no collective in the chain."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP path has no CPU fallback)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion, device_info, worlds
    B = args.graphs_per_gpu or 256
    S = args.samples_per_step
    dims = worlds.MODE_DIMS['qualitative']
    host_budget = apply_lanes_rule(world)
    den = ConstraintDiffuser(dims=dims, hidden_dim=HIDDEN, input_mode='qualitative', EBM='ULA', device=dev, verbose=False, model='StructDiffusion')
    den.reset_parameters(0)
    # nn.Linear default init everywhere except the last pose-decoder layer, scaled by 1/20: with the default there the untrained network's
    # eps has unit scale, x0 = a_t x - b_t eps blows up in the first timesteps and the rest of the chain computes on Inf / NaN.  Same
    # kernels, same flops; the poses stay finite (config.outputs_finite)
    sd = den.state_dict()
    last = sorted(k for k in sd if k.startswith('pose_decoder.'))[-2:]
    for k in last:
        sd[k] = sd[k] * 0.05
    den.load_state_dict(sd)
    gd = GaussianDiffusion(den, timesteps=T_STEPS, EBM='ULA', samples_per_step=S)
    batch_np = worlds.qualitative_batch(B, 7, seed=5 + rank)
    n_nodes = batch_np.x.shape[0]
    base = batch_np.to_torch(dev)

    def one_step(k):
        return gd.sample(base.clone(), seed=1000 + k, row_offset=rank * n_nodes)
    for k in range(args.warmup):
        one_step(k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    cpu0 = host_cpu_seconds()
    t0 = time.perf_counter()
    for k in range(args.steps):
        x = one_step(args.warmup + k)
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0
    cpu_s = host_cpu_seconds() - cpu0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    times = rank_times(local_elapsed, world, dist, dev)
    cpu_per_rank = gather_floats(cpu_s, world, dist, dev)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * B * args.steps / elapsed
    M, Wd = 8 * B, 2 * HIDDEN
    mma = os.environ.get('CCSP_MMA', 'f16x2')
    prods, pipe = (3, 'f16') if mma == 'f16x2' else (1, 'f32')
    flops_eval = 4 * 2.0 * M * 12 * Wd * Wd                      # per block: in_proj 3 Wd^2, out_proj Wd^2, c_fc 4 Wd^2, c_proj 4 Wd^2 per token
    rec = {'metric': 'samples/sec, T=1000 ULA, StructDiffusion transformer baseline, 7-obj (all chains per second; not a BASELINE.json configuration)',
           'value': value, 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
           'lanes': gd.chain_lanes(), 'host_cpu_s_per_step': [c / args.steps for c in cpu_per_rank],
           'host_cores_busy_per_rank': [c / max(1e-9, t) for c, t in zip(cpu_per_rank, times)], 'host_budget': host_budget,
           'per_rank_ms_per_step': [1e3 * v / args.steps for v in times],
           'dtype': 'f32 (f16x2 split operands: 3 fp16 MFMA products per fp32 product, fp32 accumulate)' if mma == 'f16x2' else 'f32',
           'data': 'synthetic (random-init weights: nn.Linear default init, last pose-decoder layer x 0.05 so that the chain stays finite)',
           'config': {'workload': 'StructDiffusion baseline (denoise_fn.py:391-451): %d graphs x 7 objects per GPU, 8-token sequences, width %d, 4 blocks, 2 heads, '
                                  'T=1000 ULA S=%d' % (B, Wd, S), 'name': 'sd', 'graphs_per_gpu': B, 'nodes_per_gpu': n_nodes, 'token_rows': M,
                      'evaluations_per_chain': T_STEPS * (1 + S), 'samples_per_step': S, 'gemm_mode': mma,
                      'parallelism': 'independent replicas x%d (graphs are independent; no collective)' % world,
                      'outputs_finite': bool(torch.isfinite(x).all().item())}}
    if rank == 0 and not args.no_roofline:
        b = base.clone()
        gd.profile(b, True)
        gd.sample(b, seed=77)
        ks = gd.kernel_stats()
        ev = ks.get('StructDiffusion evaluation')
        if ev:
            us = 1e3 * ev[1]
            rec['roofline'] = {'bound': 'mfma', 'achieved': prods * flops_eval / (us * 1e-6) / 1e12, 'peak': PEAKS[pipe], 'unit': 'TFLOP/s',
                               'frac': prods * flops_eval / (us * 1e-6) / 1e12 / PEAKS[pipe], 'traffic': None,
                               'kernel': 'one whole evaluation (26 kernel launches: embed + ln_1, 4 x [in_proj, attention, out_proj, c_fc, c_proj, ln_2 + next ln_1], ln_2 + decode, node update); '
                                         'per-kernel durations: profiles/r06_kernel_stats_sd.csv',
                               'us_per_evaluation': us, 'calls_timed': ev[0], 'executed_flops_fp32_equiv_per_evaluation': flops_eval,
                               'products_per_fp32_product': prods, 'pipe': pipe,
                               'frac_fp32_equiv': flops_eval / (us * 1e-6) / 1e12 / PEAKS['f32'],
                               'note': 'flops of the four GEMMs of the four blocks (attention scores, LayerNorms, decoder not counted) x MFMA products per fp32 product / '
                                       'the mean duration of a whole evaluation (HIP events on the chain stream) / the dense peak of the pipe'}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # round 6: the same KIND of baseline as c2 / c4 / c5 -- the cost-faithful PyTorch-CPU port of the reference's transformer path
        # (oracle/torch_proxy.py ProxyStructDiffuser; certified against the imported reference: outputs bit-identical, cost per evaluation within
        # +-5 %, profiles/r06_certify_proxy.txt) timed on this box's host cores on whole timesteps of the same 256-graph batch
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import torch_proxy                              # the baseline port, never the product path
        cpu_batch = batch_np.to_torch('cpu')
        cpu_batch.num_graphs = B
        r = torch_proxy.time_baseline({k: v.cpu().numpy() for k, v in den.state_dict().items()}, dims, HIDDEN, 13, cpu_batch, T=T_STEPS, S=S,
                                      n_timesteps=3, budget_s=25.0, sampler='ULA', model_kind='StructDiffusion')
        rec['cpu_baseline'] = {'value': r['samples_per_s'], 'unit': 'samples/s', 'cores': r['cores'], 'threads_used': r['cores'], 'host_cpus': os.cpu_count(),
                               'host_cpu_quota': effective_cores(), 'kind': 'port', 'sample': r['sample'], 'sec_per_timestep': r['sec_per_timestep'],
                               'sec_per_eval_by_threads': r['sec_per_eval_by_threads'], 'speedup_gpu_over_cpu': value / r['samples_per_s']}
    if rank == 0:
        rec['device'] = device_info()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(rec), flush=True)



def effective_cores():
    """cores this container may actually use: the cgroup quota if there is one (the GPU boxes report 256 CPUs and grant 16), else os.cpu_count()"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n

def host_cpu_seconds():
    """user + system CPU seconds of this process so far (getrusage(RUSAGE_SELF): the library's lane threads, RCCL's proxy threads and the Python
    thread included; child processes -- the rocprofv3 sub-runs -- are not)"""
    import resource
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime + r.ru_stime


# What a rank costs its host, measured (profiles/r06_host_budget.txt, tools/host_budget.sh; C2, one MI355X box with a 16-core cgroup quota): 2.67 busy
# cores with two lanes (the two enqueueing threads ~0.8 each -- a chain is 33 000 launches per lane -- plus the HIP runtime's own threads), 1.9 with one
# lane, which runs the chain 8 % slower (465 against 506 samples/s).  With K busy competitor processes in the same cgroup (= the other ranks of an
# N-rank run, 2.67 or 1.9 each): two lanes hold 507 up to K = 13, 499 at 16, 450 at 19 (seven other two-lane ranks), 386 at 24; one lane 466 up to 16,
# 444 at 19.  So on 16 cores two lanes win up to 7 ranks (499 against 466) and lose at 8 (450 against the 465 that eight ONE-lane ranks get with 13
# competitors each): the launcher takes one lane per rank below 2.2 usable cores per rank.  The reference is ONE process for the whole node
# (ddpm.py:342-351), so this cost is the port's own -- and every line reports it (host_cpu_s_per_step, host_cores_busy_per_rank, lanes).
CORES_PER_RANK_TWO_LANES = 2.2


def select_lanes(world, cores, env_lanes=None):
    """-> (value for CCSP_LANES or None = the library's default of two lanes, reason).  An explicit CCSP_LANES always wins; otherwise one lane per rank
    when the container grants fewer than CORES_PER_RANK_TWO_LANES cores per rank (8 ranks on the GPU boxes' 16-core quota: 16 < 17.6 -> one lane)."""
    if env_lanes not in (None, ''):
        return None, 'CCSP_LANES=%s set by the caller' % env_lanes
    if cores < CORES_PER_RANK_TWO_LANES * world:
        return 1, ('one lane per rank: %d usable cores < %.1f x %d ranks (a two-lane rank keeps 2.67 host cores busy, a one-lane rank 1.9: profiles/r06_host_budget.txt)' % (cores, CORES_PER_RANK_TWO_LANES, world))
    return None, 'library default (two lanes above 6144 active edges): %d usable cores >= %.1f x %d ranks' % (cores, CORES_PER_RANK_TWO_LANES, world)


def apply_lanes_rule(world):
    """decide CCSP_LANES for this rank BEFORE the model is created (the library reads it at ccsp_model_create); -> dict for the JSON line"""
    cores = int(os.environ.get('CCSP_BENCH_HOST_CORES', '0') or 0) or effective_cores()      # (the override: tests of the rule, tools/host_budget.sh)
    lanes, why = select_lanes(world, cores, os.environ.get('CCSP_LANES'))
    if lanes is not None:
        os.environ['CCSP_LANES'] = str(lanes)
    return {'usable_cores': cores, 'cores_per_rank_needed_for_two_lanes': CORES_PER_RANK_TWO_LANES, 'ccsp_lanes_env': os.environ.get('CCSP_LANES'), 'rule': why}


def gather_floats(v, world, dist, dev):
    """one double per rank, on every rank"""
    if dist is None or world == 1:
        return [float(v)]
    t = torch.tensor([v], device=dev, dtype=torch.float64)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return [float(q.item()) for q in parts]


def world_label(cfg):
    """'RandomSplitQualitativeWorld 8-obj' from the configuration's label"""
    return re.sub(r' \d+ objects$', '', cfg['label'].split(':')[1].split(',')[0].strip()) + ' %d-obj' % cfg['n_objects']


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def self_launch(n):
    """`python bench.py --gpus N` with WORLD_SIZE unset: become N ranks -- this same command line under torch.distributed.run, one process per
    GPU, rendezvous on 127.0.0.1 at a free port.  The ranks' stdout passes through (rank 0 prints the one JSON line); -> exit code"""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // max(1, n))))      # torchrun would set 1: the lanes' host threads need a few
    print('bench.py: launching %d ranks: %s' % (n, ' '.join(cmd)), file=sys.stderr, flush=True)
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit(rc)
    return 0


def rank_times(elapsed, world, dist, dev):
    """every rank's wall time of the timed region, on every rank (one all_gather of a double)"""
    if dist is None or world == 1:
        return [float(elapsed)]
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def communicator_proof(dist, dev, backend):
    """what shows that the collective library joined N ranks: the process group's size, a sum over ranks whose value only N distinct
    participants produce ([1, rank] -> [N, N(N-1)/2]), and -- on RCCL -- ncclCommCount of a communicator the LIBRARY creates over the same
    ranks (ccsp_rccl_comm_create; the one the MALA global-batch reduction uses) with the same sum through ncclAllReduce from C"""
    world, rank = dist.get_world_size(), dist.get_rank()
    t = torch.tensor([1.0, float(rank)], device=dev, dtype=torch.float32)
    dist.all_reduce(t)
    proof = {'backend': backend, 'world': world, 'allreduce_of_[1,rank]': [float(v) for v in t.cpu()],
             'allreduce_expected': [float(world), float(world * (world - 1) // 2)]}
    if backend == 'nccl':
        # (the proof must never cost the line: the timed region is over, the measurement stands on torch's process group alone -- a failure of the
        #  library-side communicator, the same on every rank (RCCL not found by dlopen, an ABI the check refuses), is reported in the line instead)
        try:
            import ctypes as C
            from diffusion_ccsp_amd import _lib, sharding
            L = _lib.lib()
            comm = sharding._native_comm(dist, dev)
            n, ver = C.c_int32(), C.c_int32()
            _lib.check(L.ccsp_rccl_comm_count(C.c_void_p(comm), C.byref(n), C.byref(ver)))
            u = torch.tensor([1.0, float(rank)], device=dev, dtype=torch.float32)
            _lib.check(L.ccsp_rccl_allreduce_sum_f32(C.c_void_p(comm), C.c_void_p(u.data_ptr()), 2, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            torch.cuda.synchronize(dev)
            proof.update({'rccl_ranks': int(n.value), 'rccl_version_code': int(ver.value), 'rccl_allreduce_of_[1,rank]': [float(v) for v in u.cpu()]})
            L.ccsp_rccl_comm_destroy(C.c_void_p(comm))
        except Exception as e:      # noqa: BLE001
            proof['rccl_ranks'] = None
            proof['rccl_error'] = '%s: %s' % (type(e).__name__, str(e)[:300])
    return proof


def dry_run_main(args):
    """the N > 1 host path on CPU (gloo or, on a GPU box, nccl): everything bench.py does around the chains -- rank bookkeeping, the weight
    file read on rank 0 and broadcast as ONE flat buffer, every rank's shard of the global batch and its row offset, the gather of the
    final poses, barrier-bracketed timing with the maximum over ranks, the JSON line -- with NO sampling (zeros stand in for the poses;
    there is no CPU fallback to run instead).  value is null.  tests/test_bench_launcher.py drives it with world size 2 every round."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    import torch.distributed as dist
    dev = torch.device('cpu')
    if args.backend == 'nccl':
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group(args.backend, rank=rank, world_size=world)
    from diffusion_ccsp_amd import sharding, worlds
    host_budget = apply_lanes_rule(world)
    cname = 'c2' if args.config in ('c3', 'sd') else args.config
    cfg = CONFIGS[cname]
    B = args.graphs_per_gpu or cfg['graphs']
    if args.scaling == 'strong':
        B = 8 * cfg['graphs'] // world
    P = worlds.MODE_DIMS[cfg['mode']][-1][0]
    batch_np = getattr(worlds, cfg['batch'])(B, cfg['n_objects'], seed=5 + rank)
    n_nodes = batch_np.x.shape[0]
    wrel = next(w for w in cfg['weights'] if os.path.isfile(os.path.join(ROOT, w)))
    ref = load_weights(os.path.join(ROOT, wrel))
    shapes = {k[:-7]: v.shape for k, v in ref.items() if k.endswith('.weight')}
    sd = sharding.broadcast_state_dict(ref if rank == 0 else None, shapes, dev, dist)
    weights_equal = all(np.array_equal(sd[k].cpu().numpy(), ref[k]) for k in sd)
    proof = communicator_proof(dist, dev, args.backend)
    sizes = [None] * world
    dist.all_gather_object(sizes, n_nodes)

    def one_step(k):
        x = torch.zeros(n_nodes, P, dtype=torch.float32, device=dev)          # NOT a sample: the chain is what the dry run leaves out
        x[:, 0] = float(rank)
        return sharding.gather_poses(x, dist, sizes)
    for k in range(args.warmup):
        one_step(k)
    dist.barrier()
    cpu0 = host_cpu_seconds()
    t0 = time.perf_counter()
    for k in range(args.steps):
        full = one_step(args.warmup + k)
    local = time.perf_counter() - t0
    cpu_per_rank = gather_floats(host_cpu_seconds() - cpu0, world, dist, dev)
    dist.barrier()
    elapsed = time.perf_counter() - t0
    times = rank_times(local, world, dist, dev)
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    owner = full[:, 0].cpu().numpy()
    gathered_ok = bool(full.shape[0] == sum(sizes) and np.array_equal(owner, np.concatenate([np.full(n, r, np.float32) for r, n in enumerate(sizes)])))
    rec = {'metric': 'samples/sec (all chains), T=1000 %s, %s' % (cfg['EBM'], world_label(cfg)),
           'value': None, 'unit': 'samples/s', 'dry_run': True, 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': 1e3 * elapsed / max(1, args.steps), 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
           'dtype': 'none (dry run: no kernels)', 'data': 'synthetic',
           'world': world, 'rccl_ranks': proof.get('rccl_ranks'), 'communicator': proof,
           'per_rank_ms_per_step': [1e3 * v / max(1, args.steps) for v in times],
           'lanes': None, 'host_cpu_s_per_step': [c / max(1, args.steps) for c in cpu_per_rank], 'host_budget': host_budget,
           'config': {'workload': 'DRY RUN of the N > 1 host path of: %s, %d graphs per GPU' % (cfg['label'], B), 'name': args.config, 'graphs_per_gpu': B,
                      'nodes_per_rank': sizes, 'weights': wrel, 'weights_broadcast_equal_to_file_on_every_rank': None,
                      'gathered_rows_in_rank_order': gathered_ok,
                      'parallelism': 'independent graph shards x%d, %s weight broadcast + final gather only' % (world, args.backend)}}
    ok = torch.tensor([int(weights_equal)], dtype=torch.int64, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    rec['config']['weights_broadcast_equal_to_file_on_every_rank'] = bool(ok.item())
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(rec), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS) + ['c3', 'sd'])
    ap.add_argument('--graphs-per-gpu', type=int, default=0, help='override the configuration\'s per-GPU shard size')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='weak: the configuration\'s shard on every GPU (C3 at N = 8).  strong: C3 proper -- 2048 graphs in total (8 shards '
                         'of the configuration), split over the N ranks')
    ap.add_argument('--samples-per-step', type=int, default=S_LANGEVIN,
                    help='Langevin steps per timestep.  Default 10 = GaussianDiffusion\'s and get_args\' default (train_utils.py:89), the setting SURVEY 8 '
                         'and BASELINE.json quote (11 000 evaluations per sample); 3 = what the reference\'s documented command line passes '
                         '(train_ddpm.py:31-35, README.md:96-100)')
    ap.add_argument('--no-throughput-regime', action='store_true', help='skip the 1024-graph sub-block of the roofline (c2)')
    ap.add_argument('--no-live-pmc', action='store_true', help='do not run the rocprofv3 --pmc passes (roofline.traffic from profiles/ instead)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-evaluate', action='store_true', help='skip the Trainer.evaluate-style solved accounting (c2 only)')
    ap.add_argument('--force-dist', action='store_true', help='initialise RCCL even for one rank (exercises the N>1 code path)')
    ap.add_argument('--mala-global-batch', action='store_true',
                    help='c4 with N > 1: couple the shards through the reference\'s batch-scalar energies (2-float all_reduce per inner step)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='torch.distributed backend of an N > 1 run (nccl = RCCL; gloo only with --dry-run)')
    ap.add_argument('--dry-run', action='store_true',
                    help='host path only, no GPU: launcher -> init_process_group -> flat weight broadcast -> gather -> JSON line with "dry_run": true '
                         'and value null (nothing is sampled: the HIP path has no CPU fallback)')
    ap.add_argument('--no-strict-fp32', action='store_true', help='skip the CCSP_MMA=f32 sub-run behind value_strict_fp32 (c2)')
    args = ap.parse_args()
    if args.backend == 'gloo' and not args.dry_run:
        raise SystemExit('--backend gloo is the CPU dry run of the launcher (add --dry-run); sampling needs a GPU and RCCL')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args.gpus)
    if args.dry_run:
        return dry_run_main(args)
    if args.config == 'sd':
        return sd_main(args)
    cname = 'c2' if args.config == 'c3' else args.config
    cfg = CONFIGS[cname]
    S = args.samples_per_step

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python bench.py --gpus N launches them itself when WORLD_SIZE is unset)'
                         % (args.gpus, world))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP path has no CPU fallback)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from diffusion_ccsp_amd import (ComposedEBMDenoiseFn, ConstraintDiffuser, GaussianDiffusion, device_info, sharding, worlds)
    from diffusion_ccsp_amd import _lib
    host_budget = apply_lanes_rule(world)           # (CCSP_LANES is read by ccsp_model_create: decided here, per rank, before any model exists)
    B = args.graphs_per_gpu or cfg['graphs']
    if args.scaling == 'strong':
        total = 8 * cfg['graphs']
        if total % world:
            raise SystemExit('--scaling strong: %d graphs do not split over %d ranks' % (total, world))
        B = total // world
    mode, energy = cfg['mode'], cfg['energy']
    dims = worlds.MODE_DIMS[mode]
    P, grasp = dims[-1][0], len(dims) == 3
    # independent shards: rank r owns graphs [r*B, (r+1)*B) of the global batch (seed 5 + rank)
    batch_np = getattr(worlds, cfg['batch'])(B, cfg['n_objects'], seed=5 + rank)
    n_nodes, n_edges = batch_np.x.shape[0], batch_np.edge_index.shape[1]

    # weights: rank 0 reads the file, every other rank receives them over RCCL (xGMI)
    wrel = next(w for w in cfg['weights'] if os.path.isfile(os.path.join(ROOT, w)))
    wpath = os.path.join(ROOT, wrel)
    den = ConstraintDiffuser(dims=dims, hidden_dim=HIDDEN, input_mode=mode, EBM=cfg['EBM'], energy_wrapper=energy, device=dev, verbose=False)
    sd = load_weights(wpath) if rank == 0 else None
    sd = sharding.broadcast_state_dict(sd, den.shapes(), dev, dist)
    den.load_state_dict(sd)
    fn = ComposedEBMDenoiseFn(den) if energy else den
    gd = GaussianDiffusion(fn, timesteps=T_STEPS, EBM=cfg['EBM'], samples_per_step=S)
    if args.mala_global_batch and dist is not None and cfg['EBM'] == 'MALA':
        sharding.enable_global_batch_energy(gd, dist)
    base = batch_np.to_torch(dev)

    def one_step(k):
        b = base.clone()                       # a fresh batch object: graph planning/upload is inside the step
        x = gd.sample(b, seed=1000 + k, row_offset=rank * n_nodes)
        if dist is not None:
            sharding.gather_poses(x, dist)
        return x

    for k in range(args.warmup):
        one_step(k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    cpu0 = host_cpu_seconds()
    t0 = time.perf_counter()
    for k in range(args.steps):
        x = one_step(args.warmup + k)
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0          # this rank's own K steps (before it waits for the others)
    cpu_s = host_cpu_seconds() - cpu0                 # user + sys of this rank over its own K steps: the lane threads' enqueue work is in here
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    times = rank_times(local_elapsed, world, dist, dev)
    cpu_per_rank = gather_floats(cpu_s, world, dist, dev)
    lanes_used = gd.chain_lanes()
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    proof = communicator_proof(dist, dev, 'nccl') if dist is not None else None
    finite = bool(torch.isfinite(x).all().item())
    nan_graphs = len(set(batch_np.batch[(~torch.isfinite(x).all(dim=1)).cpu().numpy()].tolist()))
    samples = world * B * args.steps
    value = samples / elapsed
    evals_per_chain = T_STEPS * (1 + (2 if cfg['EBM'] == 'MALA' else 1) * S)

    rec = {
        # `value` is ALL chains per second, and the metric says so; BASELINE.json's "solved samples/sec" is the separate top-level pair
        # solved_metric / solved_samples_per_s (= value x solved_fraction; c2 / c3 only: the other worlds have no solved check)
        'metric': 'samples/sec (all chains), T=1000 %s, %s' % (cfg['EBM'], world_label(cfg)),
        'value_note': ('value = all chains per second; BASELINE.json\'s metric proper is solved_samples_per_s (its name: solved_metric)' if args.config in ('c2', 'c3') else
                       'value = all chains per second with the gradient evaluations of unmoved states skipped, a property of the acceptance rate; '
                       'value_recomputing_every_evaluation and mean_acceptance_rate beside it at top level; no solved check for this world' if cfg['EBM'] == 'MALA' else
                       'value = all chains per second; no solved check for this world'),
        'value': value, 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': args.scaling,
        'vs_baseline': None,
        'value_per_gpu': value / world, 'world': world, 'rccl_ranks': (proof or {}).get('rccl_ranks'),
        'per_rank_ms_per_step': [1e3 * v / args.steps for v in times],
        'per_rank_ms_per_step_min_max': [1e3 * min(times) / args.steps, 1e3 * max(times) / args.steps],
        # what a rank costs its host: CPU-seconds (user + sys, every thread of the process) per step and per second of its own chain time.  A rank whose
        # busy-cores figure falls below what one rank alone shows (profiles/r06_host_budget.txt) while its ms_per_step rises is starved of host cores.
        'lanes': lanes_used,
        'host_cpu_s_per_step': [c / args.steps for c in cpu_per_rank],
        'host_cores_busy_per_rank': [c / max(1e-9, t) for c, t in zip(cpu_per_rank, times)],
        'host_budget': host_budget,
        'dtype': {'f16x2': 'f32 (f16x2 split operands: 3 fp16 MFMA products per fp32 product, fp32 accumulate)',
                  'bf16x3': 'f32 (bf16x3 split operands: 6 bf16 MFMA products per fp32 product, fp32 accumulate)'}.get(os.environ.get('CCSP_MMA', 'f16x2'), 'f32'),
        'data': 'synthetic',
        'config': {'workload': '%s, %d graphs per GPU, hidden_dim %d' % (cfg['label'] if S == S_LANGEVIN else cfg['label'].replace('S=10', 'S=%d (NOT the headline setting: '
                               'the reference\'s documented command line, train_ddpm.py:31-35)' % S), B, HIDDEN),
                   'samples_per_step': S,
                   'name': args.config, 'graphs_per_gpu': B, 'nodes_per_gpu': n_nodes, 'edges_per_gpu': n_edges,
                   'evaluations_per_chain': evals_per_chain,
                   'parallelism': 'independent graph shards x%d, RCCL weight broadcast + final gather only' % world +
                                  (' + MALA global-batch energies (2-float all_reduce per inner step)' if args.mala_global_batch and dist is not None and cfg['EBM'] == 'MALA' else ''),
                   'gemm_mode': os.environ.get('CCSP_MMA', 'f16x2'),
                   'weights': wrel, 'weights_storage': weights_storage(wpath), 'weights_note': WEIGHT_NOTES.get(wrel, ''),
                   'outputs_finite': finite, 'graphs_with_nonfinite_poses': nan_graphs},
    }

    replica_value = None
    if args.mala_global_batch and dist is not None and cfg['EBM'] == 'MALA':
        # the same shards WITHOUT the coupling (replica semantics: each shard its own reference batch, no collective in the chain),
        # timed the same way on every rank, so that the line shows what the 10 000 two-float all-reduces per chain cost
        sharding.enable_global_batch_energy(gd, None)
        one_step(0)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        one_step(args.warmup + args.steps)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        tt = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        replica_value = world * B / float(tt.item())
        sharding.enable_global_batch_energy(gd, dist)

    if cfg['EBM'] == 'MALA' and rank == 0:
        # MALA: the gradient evaluation of an inner step whose predecessor accepted no node is skipped on the device (exact: the
        # state has not moved; include/ccsp.h, ccsp_chain_skipped).  Report what that was worth: the acceptance of the last timed
        # chain, the evaluations it skipped, and the same chain WITH every evaluation recomputed (one extra untimed-region chain).
        st = gd.chain_stats()
        acc = gd.last_accept_rates
        reuse_on = os.environ.get('CCSP_MALA_REUSE', '1') != '0'
        rec['mala'] = {'rejected_step_reuse': reuse_on, 'evaluations_enqueued_per_chain': int(st['evals']),
                       'evaluations_skipped_last_chain': int(st['evals_skipped']),
                       'mean_acceptance_rate': float(acc.float().mean().item()) if acc is not None else None,
                       'note': 'an inner step that accepts no node leaves x unchanged; E(x) and dE/dx of the next step are then the values already '
                               'computed, and their kernels return at once.  Bitwise the chain that recomputes '
                               '(test_mala_rejected_step_reuse_is_bitwise_identical); the gain is workload-dependent (acceptance rate).  The profiled chain of the '
                               'roofline block recomputes every evaluation (kernels timed at full work).'}
        rec['mean_acceptance_rate'] = rec['mala']['mean_acceptance_rate']
        if replica_value is not None:
            rec['mala']['value_global_batch'] = value
            rec['mala']['value_replica_semantics'] = replica_value
            rec['mala']['global_batch_reduction'] = ('ncclAllReduce(sum, 2 floats) per inner step, enqueued by the library on the chain stream '
                                                     '(ccsp_model_set_energy_allreduce; communicator of its own over the %d ranks)' % world)
        if reuse_on and dist is None:
            os.environ['CCSP_MALA_REUSE'] = '0'
            den0 = ConstraintDiffuser(dims=worlds.MODE_DIMS[cfg['mode']], hidden_dim=HIDDEN, input_mode=cfg['mode'], EBM=cfg['EBM'],
                                      energy_wrapper=cfg['energy'], device=dev, verbose=False)
            den0.load_state_dict(sd)
            gd0 = GaussianDiffusion(ComposedEBMDenoiseFn(den0) if cfg['energy'] else den0, timesteps=T_STEPS, EBM=cfg['EBM'], samples_per_step=S)
            os.environ['CCSP_MALA_REUSE'] = '1'
            gd0.sample(base.clone(), seed=999)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            x0 = gd0.sample(base.clone(), seed=1000 + args.warmup + args.steps - 1, row_offset=rank * n_nodes)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            rec['mala']['value_recomputing_every_evaluation'] = B / dt
            rec['value_recomputing_every_evaluation'] = B / dt          # (the kernels-at-full-work number, next to `value`)
            rec['mala']['recomputing_chain_bitwise_equal'] = bool(torch.equal(x0, x) or ((x0 == x) | (torch.isnan(x0) & torch.isnan(x))).all().item())

    if cname == 'c2':
        # "solved?" check (diffusion-ccsp_amd/checker.py); outside the timed region
        from diffusion_ccsp_amd import checker, evaluate
        solved = checker.solved_mask(x.detach().cpu().numpy(), batch_np)
        n_solved = torch.tensor([int(solved.sum()), int(solved.size)], device=dev, dtype=torch.int64)
        if dist is not None:
            dist.all_reduce(n_solved)
        solved_fraction = float(n_solved[0].item()) / max(1, int(n_solved[1].item()))
        rec['solved_fraction'] = solved_fraction
        rec['solved_metric'] = 'solved samples/sec, T=1000 ULA, RandomSplitQualitativeWorld 8-obj (BASELINE.json metric)'
        rec['solved_samples_per_s'] = value * solved_fraction
        # BASELINE.json's metric under BASELINE.json's name, with what bounds it (the reader gets value / value_strict_fp32 / this from `parsed` alone)
        rec['baseline_metric'] = {
            'name': 'solved samples/sec, T=1000 ULA, RandomSplitQualitativeWorld 8-obj',
            'value': value * solved_fraction, 'unit': 'solved samples/s', 'solved_fraction': solved_fraction,
            'solved_of': [int(n_solved[0].item()), int(n_solved[1].item())],
            'arithmetic': 'the headline mode (see dtype); one try per graph, no rejection sampling',
            'bounded_by': 'weights, not the sampler port: no reference checkpoint exists offline, the committed weights are the reference recipe (trained on 2-5 objects) at '
                          '30 000 steps, and the REFERENCE sampler with exactly these weights solves 0 of 16 8-object graphs (tests/golden/chain_q256_bench_B16, '
                          'test_bench_weights_vs_reference_golden: same final poses, same solved mask)'}
        rec['config']['solved_note'] = ('solved_fraction = share of the last timed batch (one try per graph, no rejection) passing the collision + '
                                        'qualitative-constraint check of diffusion-ccsp_amd/checker.py; the reference sampler itself overflows fp32 in its '
                                        'first timesteps on some graphs (ULA step 2*beta with beta -> 0.999; Trainer.evaluate skips them, ddpm.py:644), see DESIGN.md section 7')
        if rank == 0 and not args.no_evaluate:
            # the reference's own accounting (Trainer.evaluate, ddpm.py:591-603,823-836): tries=(10, 0), top-1 / top-10
            import tempfile
            rng = np.random.default_rng(11)
            sets = {}
            for n_obj in (3, 8):
                gs = []
                for _ in range(100):
                    wd = worlds.sample_qualitative_world(rng, n_obj)
                    gs.append(worlds.encode_qualitative(wd['nodes'], wd['constraints']))
                sets[n_obj] = gs
            with tempfile.TemporaryDirectory() as td:
                log = evaluate.Evaluator(gd, sets, td).evaluate(0, tries=(10, 0), run_all=True, seed=500)
            rec['evaluate'] = {'pattern': 'Trainer.evaluate: test sets of 100 graphs, tries=(10, 0), batch size 100 (ddpm.py:591-603)',
                               'sets': {k: {'success_rate': v['success_rate'], 'success_rate_top10': v.get('success_rate_top10', v.get('success_rate_top3')),
                                            'sampling_s_per_graph': float(np.mean([s[2] for s in v['sampling_time']]))} for k, v in log.items()}}

    if proof is not None:
        rec['communicator'] = proof

    if cname == 'c2' and world == 1 and os.environ.get('CCSP_MMA', 'f16x2') == 'f16x2' and not args.no_strict_fp32:
        # the same step with every GEMM on the fp32 matrix pipe (CCSP_MMA=f32: v_mfma_f32_32x32x2_f32, no operand split): a second model of
        # the same weights, 1 warm-up + 3 timed steps bracketed like the headline.  Read at model creation, so the variable is set around it.
        os.environ['CCSP_MMA'] = 'f32'
        try:
            den32 = ConstraintDiffuser(dims=dims, hidden_dim=HIDDEN, input_mode=mode, EBM=cfg['EBM'], energy_wrapper=energy, device=dev, verbose=False)
            den32.load_state_dict(sd)
            gd32 = GaussianDiffusion(den32, timesteps=T_STEPS, EBM=cfg['EBM'], samples_per_step=S)
        finally:
            del os.environ['CCSP_MMA']
        gd32.sample(base.clone(), seed=999, row_offset=rank * n_nodes)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(3):
            x32 = gd32.sample(base.clone(), seed=1000 + args.warmup + args.steps - 3 + k, row_offset=rank * n_nodes)
        torch.cuda.synchronize()
        rec['value_strict_fp32'] = 3 * B / (time.perf_counter() - t1)
        ok32 = torch.isfinite(x32).all(dim=1) & torch.isfinite(x).all(dim=1)
        rec['strict_fp32'] = {'gemm_mode': 'f32', 'steps': 3, 'warmup': 1,
                              'max_abs_diff_final_poses_vs_headline_mode': float((x32 - x)[ok32].abs().max().item()) if bool(ok32.any().item()) else None,
                              'note': 'value_strict_fp32 = the same step with CCSP_MMA=f32 (fp32 MFMA, no fp16 operand split); the last of its chains has the '
                                      'seed of the last headline chain'}
        del gd32, den32

    if rank == 0 and not args.no_roofline:
        # separate profiled pass: a HIP event before every launch of one more chain, on the stream the kernels run on
        b = base.clone()
        gd.profile(b, True)
        gd.sample(b, seed=77)
        st = gd.chain_stats()
        ks = gd.kernel_stats()
        plan = _lib.plan_host(n_nodes, cfg['n_types'], batch_np.edge_index, batch_np.edge_attr)
        R, E_act = plan['R'], plan['E_act']
        mma = os.environ.get('CCSP_MMA', 'f16x2')
        f_eval, f_node, f_edge = algorithmic_flops(n_nodes, E_act, HIDDEN, P, grasp)
        # which variants of the tile kernels this (one-lane, full-batch) pass ran, and the counters of the same variants
        row_mode, edge_tile = gd.kernel_variant()
        variant = {'row GEMM (forward)': 'k_rowgemm_h2<256, 512, %d>' % row_mode,
                   'edge decoder (forward)': 'k_edge_h2s' if edge_tile == 16 else 'k_edge_h2<%s, %d' % ('true' if energy else 'false', edge_tile // 32)}
        syms = sorted(set(filter(None, (kernel_symbol(label, mma, energy) for label in ks))))
        pmc_cfg = args.config if args.config != 'c3' else 'c2'
        traffic, stamp = ({}, None)
        rstats, rstamp = ({}, None)
        if world == 1 and not args.no_live_pmc and mma == 'f16x2':
            del b
            traffic, stamp = pmc_traffic_live(pmc_cfg, B, syms)
            rstats, rstamp = rocprof_kernel_stats_live(pmc_cfg, B, syms)
        if not traffic:
            live_reason = stamp
            traffic, stamp = pmc_traffic_file(pmc_cfg, syms)
            if stamp is not None and live_reason is not None:
                stamp['live_pass'] = live_reason.get('reason', 'not run')
        kernels = []
        for label, (calls, ms) in ks.items():
            w = executed_work(label, n_nodes, E_act, R, HIDDEN, mma)
            sym = kernel_symbol(label, mma, energy)
            ent = {'kernel': label, 'symbol': sym, 'calls_timed': calls, 'us_mean': 1e3 * ms}
            if w is not None:
                flops, prods, pipe = w
                ent.update({'executed_flops_fp32_equiv': flops, 'products_per_fp32_product': prods, 'pipe': pipe,
                            'pipe_tflops': prods * flops / (ms * 1e-3) / 1e12, 'pipe_peak_tflops': PEAKS[pipe],
                            'frac': prods * flops / (ms * 1e-3) / 1e12 / PEAKS[pipe]})
            rs = rstats.get(sym)
            if rs is not None and (variant.get(label) is None or rs['kernel_name'].startswith(variant[label])):
                ent['rocprof_avg_ns'] = rs['avg_ns']          # the profiler's AverageNs of the same kernel (separate pass; us_mean spans launch to next mark)
                ent['rocprof_calls'] = rs['calls']
                ent['rocprof_min_max_ns'] = [rs['min_ns'], rs['max_ns']]
                if w is not None:
                    ent['frac_at_rocprof_avg'] = prods * flops / (rs['avg_ns'] * 1e-9) / 1e12 / PEAKS[pipe]
            tr = traffic.get(sym)
            if tr is not None:
                want = variant.get(label)
                if want is None or tr['kernel_name'].startswith(want):
                    ent['fabric_bytes_per_launch'] = tr['bytes']
                    ent['frac_bytes'] = tr['bytes'] / (ms * 1e-3) / 8.0e12
                    ent['counters_taken_on'] = tr['kernel_name']
                else:                        # counters of another variant of the kernel: not this launch's traffic
                    ent['fabric_bytes_refused'] = 'counters are of %s, the timed pass ran %s' % (tr['kernel_name'], want)
            kernels.append(ent)
        timed = sum(k['us_mean'] * k['calls_timed'] for k in kernels)
        for k in kernels:
            k['share_of_timed'] = k['us_mean'] * k['calls_timed'] / timed if timed else None
        dom = max((k for k in kernels if 'frac' in k), key=lambda k: k['us_mean'] * k['calls_timed'])
        # evaluations seen by the profiler = launches of the forward row GEMM
        n_eval_timed = ks.get('row GEMM (forward)', (0, 0.0))[0]
        us_eval = timed / n_eval_timed if n_eval_timed else None
        types_present = int(len(set(np.asarray(batch_np.edge_attr).astype(np.int64).tolist()) & set(range(cfg['n_types']))))
        frac_mfma = dom['frac']
        frac_bytes = dom.get('frac_bytes')
        by_bytes = frac_bytes is not None and frac_bytes > frac_mfma
        eval_labels = [k for k in kernels if k['kernel'] not in ('energy sum', 'HMC elementwise')]
        if any(k['kernel'] == 'edge decoder forward + backward' for k in kernels):
            # energy mode since round 6: a GRADIENT evaluation (what fabric_bytes_per_evaluation is quoted per, and what tools/profile_eval.py runs) is
            # row GEMM -> fused decoder forward + backward -> transpose GEMM -> node gradient; the forward-only decoder launches timed above belong
            # to MALA's energy evaluation at the proposal, not to it
            eval_labels = [k for k in eval_labels if k['kernel'] != 'edge decoder (forward)']
        fabric_eval = sum(k['fabric_bytes_per_launch'] for k in eval_labels) if eval_labels and all('fabric_bytes_per_launch' in k for k in eval_labels) else None
        alg_bytes = algorithmic_bytes(n_nodes, types_present, HIDDEN, P, grasp)
        # what one evaluation touches: if it fits the 256 MiB Infinity Cache the fabric-side bytes (L2 misses, Infinity-Cache hits
        # included) are NOT HBM bytes and the byte fraction is a fabric-side figure, not an HBM roofline
        ws = working_set_bytes(n_nodes, E_act, R, HIDDEN, P, types_present, energy)
        in_cache = ws < 256 * 1024 * 1024
        hbm_bound = by_bytes and not in_cache
        rec['roofline'] = {
            'bound': 'hbm' if hbm_bound else ('fabric' if by_bytes else 'mfma'),
            'achieved': (dom['fabric_bytes_per_launch'] / (dom['us_mean'] * 1e-6) / 1e9) if hbm_bound else dom['pipe_tflops'],
            'peak': 8000.0 if hbm_bound else dom['pipe_peak_tflops'], 'unit': 'GB/s' if hbm_bound else 'TFLOP/s',
            'frac': frac_bytes if hbm_bound else frac_mfma,
            'frac_mfma': frac_mfma, 'frac_bytes': frac_bytes,
            'working_set_bytes': ws, 'working_set_fits_infinity_cache': in_cache,
            'bound_note': 'bound = "fabric": the dominant kernel\'s fabric-side bytes / duration / 8 TB/s (frac_bytes) exceed its matrix-pipe fraction, but one '
                          'evaluation\'s working set fits the 256 MiB Infinity Cache, so those bytes are L2 misses served on the die, not HBM traffic: '
                          'achieved / peak / frac are the MATRIX-PIPE figures (frac = frac_mfma).  bound = "hbm" only when the working set exceeds the '
                          'Infinity Cache (see throughput_regime); "mfma" when the pipe fraction is the larger one',
            'traffic': dom.get('fabric_bytes_per_launch'), 'traffic_source': stamp,
            'rocprof_avg_ns': dom.get('rocprof_avg_ns'), 'frac_at_rocprof_avg': dom.get('frac_at_rocprof_avg'), 'rocprof_source': rstamp,
            'fabric_bytes_per_evaluation': fabric_eval, 'algorithmic_bytes_per_evaluation': alg_bytes,
            'wasted_traffic_ratio': (fabric_eval / alg_bytes) if fabric_eval else None,
            'kernel': '%s: %s' % (dom['kernel'], dom.get('counters_taken_on') or variant.get(dom['kernel']) or dom['symbol']),
            'note': 'the dominant kernel priced twice: frac_mfma = flops it EXECUTES on the %s matrix pipe (%d MFMA products per fp32 product after '
                    'the row factorisation) / its mean launch duration (HIP events on the chain stream, launch to next mark) / the dense %s peak; '
                    'frac_bytes = its fabric-side bytes (2 x FETCH_SIZE + WRITE_SIZE: Infinity-Cache hits included, an upper bound on HBM bytes) / '
                    'the same duration / 8 TB/s' %
                    (dom['pipe'], dom['products_per_fp32_product'], dom['pipe']),
            'frac_fp32_equiv': dom['executed_flops_fp32_equiv'] / (dom['us_mean'] * 1e-6) / 1e12 / PEAKS['f32'],
            'kernels': kernels,
            'us_per_evaluation_timed': us_eval,
            'algorithmic': {'flops_per_evaluation': f_eval * (2 if energy else 1), 'flops_per_node': f_node, 'flops_per_edge': f_edge,
                            'note': 'SURVEY.md 8(d): the reference\'s dense per-edge formulation (energy mode: forward + input gradient = 2x); '
                                    'algorithmic_tflops = that / the timed evaluation, NOT a utilisation: the kernels execute %.2fx fewer flops'
                                    % (f_eval / max(1.0, sum(k.get('executed_flops_fp32_equiv', 0.0) for k in kernels if k['kernel'] in
                                                             ('row GEMM (forward)', 'edge decoder (forward)', 'node update + pose encoder')))),
                            'algorithmic_tflops': (f_eval * (2 if energy else 1) / (us_eval * 1e-6) / 1e12) if us_eval else None,
                            'whole_chain_algorithmic_tflops': f_eval * (2 if energy else 1) * st['evals'] / (st['ms_total'] * 1e-3) / 1e12 / (2 if energy else 1)},
            'chain_ms_event': st['ms_total'], 'chain_evals': st['evals'],
            'profile_note': 'the profiled chain runs as one lane with an event per launch; its ms is not the timed value above',
        }
        if 'value_strict_fp32' in rec:
            # the number measured on the reference's own arithmetic width, priced against the fp32 matrix peak over the WHOLE evaluation:
            # executed fp32 flops of one evaluation (after the row factorisation) x evaluations per chain x chains per second / 157.3 TFLOP/s
            ex = sum(k.get('executed_flops_fp32_equiv', 0.0) for k in kernels if k['kernel'] in ('row GEMM (forward)', 'edge decoder (forward)', 'node update + pose encoder'))
            rec['roofline']['value_strict_fp32'] = rec['value_strict_fp32']
            rec['roofline']['frac_fp32_peak'] = rec['value_strict_fp32'] / B * evals_per_chain * ex / 1e12 / PEAKS['f32']
            rec['roofline']['strict_fp32_note'] = ('value_strict_fp32 = the same step with CCSP_MMA=f32 (v_mfma_f32_32x32x2_f32, no operand split): samples/s on the reference\'s own '
                                                   'arithmetic; frac_fp32_peak = its executed fp32 flops per second / the dense fp32 matrix peak (%.1f TFLOP/s), whole evaluation; '
                                                   '`frac` above is the headline mode\'s dominant kernel on the f16 pipe (three fp16 products per fp32 product)' % PEAKS['f32'])
        if cname == 'c2' and world == 1 and mma == 'f16x2' and not args.no_throughput_regime and not args.graphs_per_gpu:
            try:
                rec['roofline']['throughput_regime'] = throughput_regime(gd, cfg, worlds, dev, args)
            except Exception as e:               # noqa: a sub-block must not take the benchmark line down
                rec['roofline']['throughput_regime'] = {'error': str(e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import torch_proxy                              # the checker / baseline port, never the product path
        cpu_batch = batch_np.to_torch('cpu')
        cpu_batch.num_graphs = B
        r = torch_proxy.time_baseline(load_weights(wpath), dims, HIDDEN, cfg['n_types'], cpu_batch, T=T_STEPS, S=S,
                                      n_timesteps=3, budget_s=25.0, sampler=cfg['EBM'])
        rec['cpu_baseline'] = {'value': r['samples_per_s'], 'unit': 'samples/s', 'cores': r['cores'], 'threads_used': r['cores'], 'host_cpus': os.cpu_count(),
                               'host_cpu_quota': effective_cores(), 'kind': 'port',
                               'cores_note': 'cores = threads_used = the PyTorch thread count the value was taken at, the fastest of sec_per_eval_by_threads '
                                             '(these small matrices do not scale to every core); host_cpus = os.cpu_count(), host_cpu_quota = what the '
                                             "container's cgroup grants of them",
                               'sample': r['sample'], 'sec_per_timestep': r['sec_per_timestep'],
                               'sec_per_eval_by_threads': r['sec_per_eval_by_threads'],
                               'speedup_gpu_over_cpu': value / r['samples_per_s']}
    if rank == 0:
        rec['device'] = device_info()
        try:
            rec['host_budget']['affinity_cpus_at_end'] = len(os.sched_getaffinity(0))      # (a `taskset` in front of the run shows here if it held)
        except (AttributeError, OSError):
            pass
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stderr.flush()
        try:                                     # RCCL prints its version banner through C stdio: flush it out BEFORE the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:                        # noqa
            pass
        if 'solved_samples_per_s' in rec:
            print('solved %.2f samples/s (%.1f %% of %.1f samples/s)' % (rec['solved_samples_per_s'], 100 * rec['solved_fraction'], value), file=sys.stderr)
        print(json.dumps(rec), flush=True)      # the one JSON line, after RCCL's own banner output


if __name__ == '__main__':
    main()
