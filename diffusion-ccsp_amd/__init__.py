"""MI355X-native reverse-diffusion sampling path of Diffusion-CCSP.

    from diffusion_ccsp_amd import ConstraintDiffuser, ComposedEBMDenoiseFn, GaussianDiffusion

mirror the reference classes of the same names (networks/denoise_fn.py, networks/ddpm.py) for the
sampling path and run it through libccsp_hip.so (include/ccsp.h).
"""
from . import checker, noise, sharding, transforms, worlds  # noqa: F401
from ._lib import CcspError, build, device_info  # noqa: F401
from .denoise_fn import ComposedEBMDenoiseFn, ConstraintDiffuser  # noqa: F401
from .ddpm import GaussianDiffusion  # noqa: F401

from . import evaluate  # noqa: F401,E402

__version__ = "0.1.0"
