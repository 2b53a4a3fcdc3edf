"""pytest configuration.  `-m "not gpu"`: oracle vs golden vectors, host logic, ABI symbols.
`-m gpu`: the parity tests proper (HIP path through the C ABI vs oracle / golden vectors)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'gpu_experiments: needs a real MI355X AND the experiments build (CCSP_EXPERIMENTS=1 python -m pytest tests '
                                       '-m gpu_experiments): the variants that lost their A/Bs, kept out of the product library and of the default -m gpu run')


def pytest_collection_modifyitems(config, items):
    """the full-size parity tests wait for CPU oracle chains that full_size_oracle_prefetch (below) starts with the session: they go last, so
    that the chains run under every other GPU test instead of inside their own"""
    items.sort(key=lambda it: it.name.startswith('test_full_size'))            # (stable: nothing else moves)


import diffusion_ccsp_amd  # noqa: E402,F401
from diffusion_ccsp_amd import worlds  # noqa: E402
import oracle  # noqa: E402  (tests are allowed to use the checker)


class Batch(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


def effective_cores():
    """cores this container may actually use: the cgroup quota if there is one (the GPU boxes report 256 CPUs and grant 16), else os.cpu_count()"""
    n = os.cpu_count() or 8
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


def golden(name):
    return np.load(os.path.join(GOLD, name + '.npz'), allow_pickle=False)


def golden_meta(z):
    import ast
    return ast.literal_eval(str(z['meta']))


def golden_batch(z, prefix=''):
    import torch
    b = Batch(x=torch.from_numpy(z[prefix + 'x']), edge_index=torch.from_numpy(z[prefix + 'edge_index']),
              edge_attr=torch.from_numpy(z[prefix + 'edge_attr']), mask=torch.from_numpy(z[prefix + 'mask']))
    for extra in ('batch', 'shuffled'):                      # StructDiffusion fixtures carry the sequences
        if prefix + extra in z.files:
            setattr(b, extra, torch.from_numpy(z[prefix + extra]))
    return b


MODE_TYPES = {'qualitative': 13, 'diffuse_pairwise': 2, 'diffuse_pairwise_box': 2, 'robot_box': 2, 'stability_flat': 3}
_weights = {}


def weights(wfile):
    if wfile not in _weights:
        _weights[wfile] = oracle.load_weights(os.path.join(GOLD, wfile))
    return _weights[wfile]


def oracle_model(mode, H, wfile, T=1000, S=10, energy=False, f64=False, model='Diffusion-CCSP'):
    return oracle.OracleModel(weights(wfile), worlds.MODE_DIMS[mode], H, MODE_TYPES[mode], timesteps=T,
                              energy_wrapper=energy, samples_per_step=S, f64=f64, model=model)


def rel_err(a, b):
    """max |a-b| scaled by (1 + max|b|) -- chain histories pass through huge transients"""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (1.0 + np.abs(b).max()))


@pytest.fixture(scope='session', autouse=True)
def full_size_oracle_prefetch(request):
    """a GPU session that will run the full-size tests of test_hip_parity.py starts their oracle work (worker threads, ctypes releases the GIL) now"""
    import torch
    if torch.cuda.is_available() and any(it.name.startswith('test_full_size') for it in request.session.items):
        import test_hip_parity
        test_hip_parity.start_full_size_oracle_work(torch.device('cuda:0'))
    yield


@pytest.fixture(scope='session')
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
