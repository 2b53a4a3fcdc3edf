"""TEST INFRASTRUCTURE ONLY -- imports the *unmodified* reference (zt-yang/diffusion-ccsp,
mounted read-only at /root/reference) on PyTorch-CPU so that golden vectors can be generated
in the build container.  Nothing here travels to the GPU box as code that is executed there:
/root/reference does not exist on it.  The product path never imports this module.

Third-party modules the reference imports but this image lacks are replaced by *empty* stubs
(or, for ``jactorch.add_dim``, the one-line broadcast it is: reference use sites
networks/denoise_fn.py:328,334,397).  The reference's own files are imported as they are.
"""
import os
import sys
import types

REF = os.environ.get("CCSP_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF, "networks", "ddpm.py"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    import torch

    if "ipdb" not in sys.modules:
        _stub("ipdb", set_trace=lambda *a, **k: None)
    if "imageio" not in sys.modules:
        _stub("imageio")
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.transforms = _stub("torchvision.transforms")
        tv.utils = _stub("torchvision.utils")
    if "torch_geometric" not in sys.modules:
        class Data(object):
            def __init__(self, **kw):
                self.__dict__.update(kw)

            def clone(self):
                out = Data()
                for k, v in self.__dict__.items():
                    out.__dict__[k] = v.clone() if torch.is_tensor(v) else v
                return out

        class Dataset(object):
            pass

        class InMemoryDataset(Dataset):
            pass

        class DataLoader(object):
            def __init__(self, *a, **k):
                raise RuntimeError("torch_geometric DataLoader is a stub")

        tg = _stub("torch_geometric")
        tg.data = _stub("torch_geometric.data", Data=Data, Dataset=Dataset,
                        InMemoryDataset=InMemoryDataset)
        tg.loader = _stub("torch_geometric.loader", DataLoader=DataLoader)
    if "jactorch" not in sys.modules:
        def add_dim(t, dim, size):
            # jactorch.add_dim: insert a new axis of length `size` at `dim` (a pure broadcast)
            t = t.unsqueeze(dim)
            shape = [-1] * t.dim()
            shape[dim] = size
            return t.expand(*shape)

        jt = _stub("jactorch", add_dim=add_dim)
        jt.nn = _stub("jactorch.nn")


_mods = None


def load():
    """returns (ddpm, denoise_fn) reference modules"""
    global _mods
    if _mods is not None:
        return _mods
    if not available():
        raise RuntimeError("reference not present at %s" % REF)
    install_stubs()
    for sub in ("", "envs", "networks"):
        p = os.path.join(REF, sub) if sub else REF
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    import denoise_fn  # noqa
    import ddpm  # noqa
    _mods = (ddpm, denoise_fn)
    return _mods


def load_envs():
    """returns (builders, data_utils) reference modules (pure numpy/python)"""
    load()
    import builders  # noqa
    import data_utils  # noqa
    return builders, data_utils
