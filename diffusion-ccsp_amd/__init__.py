"""MI355X-native reverse-diffusion sampling path of Diffusion-CCSP.

    from diffusion_ccsp_amd import ConstraintDiffuser, ComposedEBMDenoiseFn, GaussianDiffusion

mirror the reference classes of the same names (networks/denoise_fn.py, networks/ddpm.py) for the
sampling path and run it through libccsp_hip.so (include/ccsp.h).
"""
import os as _os

# Kernel arguments in device memory: the dispatch of every launch reads them, and a chain is 33 000 short dependent launches.  ROCm 7
# does this by default; with HIP_FORCE_DEV_KERNARG=0 the same chain runs 13 % (C2) to 27 % (C5, C1) slower (profiles/r03_findings.md).
# Read by the HIP runtime when it initialises, i.e. at the process's first HIP call: set here only if the user has not set it and the
# runtime is not up yet.  bench.py sets it itself before it imports torch (it initialises HIP and RCCL before it imports this package).
if 'HIP_FORCE_DEV_KERNARG' not in _os.environ:
    import sys as _sys
    _t = _sys.modules.get('torch')
    if _t is not None and getattr(_t, 'cuda', None) is not None and _t.cuda.is_initialized():
        # too late to take effect in this process (and not ours to change behind a running runtime): say so instead of pretending
        import warnings as _w
        _w.warn('diffusion_ccsp_amd: the HIP runtime is already initialised and HIP_FORCE_DEV_KERNARG is unset; ROCm >= 7 keeps kernel arguments '
                'in device memory by default -- on an older default export HIP_FORCE_DEV_KERNARG=1 before the first HIP call (13-27 % on chains)')
    else:
        _os.environ['HIP_FORCE_DEV_KERNARG'] = '1'

from . import checker, noise, sharding, transforms, worlds  # noqa: F401,E402
from ._lib import CcspError, build, device_info  # noqa: F401,E402
from .denoise_fn import ComposedEBMDenoiseFn, ConstraintDiffuser  # noqa: F401,E402
from .ddpm import GaussianDiffusion  # noqa: F401,E402

from . import evaluate  # noqa: F401,E402

__version__ = "0.1.0"
