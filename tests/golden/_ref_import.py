"""Import the reference's hot-path modules in THIS container, under stubs.

Used ONLY by ``tests/golden/make_golden.py`` to capture golden vectors from the
reference's own functions.  Nothing here (and nothing from ``/root/reference``)
is needed at test time or on the GPU box: the captured ``.npz`` / ``.json``
fixtures next to this file are what the tests read.

The reference needs eight third-party packages that are not installed here
(SURVEY.md section 8c).  None of them takes part in the arithmetic of the hot
path, so each one is replaced by an empty module that only carries the names
the reference's ``import`` statements look up.
"""
import os
import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = "/root/reference"


class _AttrDict(dict):
    """Minimal attribute dict standing in for ``easydict.EasyDict``."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _AttrDict):
            v = _AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class _LightningModule(nn.Module):
    """No-op stand-in: the hot path only uses ``self.log`` and ``self.trainer``."""

    def __init__(self):
        super().__init__()
        self.logged = {}

    def log(self, name, value, *a, **k):
        self.logged[name] = value


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    sys.dont_write_bytecode = True  # the reference tree is read-only
    os.environ.setdefault("BASE_PATH", REFERENCE_ROOT)
    os.environ.setdefault("DATA_PATH", "/tmp/peclr_no_data")
    os.environ.setdefault("SAVED_MODELS_BASE_PATH", "/tmp/peclr_no_models")
    os.environ.setdefault("SAVED_META_INFO_PATH", "/tmp/peclr_no_meta")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    _mod("easydict", EasyDict=_AttrDict)
    _mod("kornia")
    _mod("comet_ml", Experiment=type("Experiment", (), {}))
    tv = _mod("torchvision")
    tv.models = _mod("torchvision.models", ResNet=type("ResNet", (nn.Module,), {}))
    tv.transforms = _mod("torchvision.transforms")
    _mod("cv2")
    pl = _mod("pytorch_lightning")
    pl.core = _mod("pytorch_lightning.core")
    pl.core.lightning = _mod(
        "pytorch_lightning.core.lightning", LightningModule=_LightningModule
    )
    pl.loggers = _mod("pytorch_lightning.loggers", comet=types.ModuleType("comet"))
    plb = _mod("pl_bolts")
    plb.optimizers = _mod("pl_bolts.optimizers")
    _mod("pl_bolts.optimizers.lars_scheduling", LARSWrapper=object)
    _mod("pl_bolts.optimizers.lr_scheduler", LinearWarmupCosineAnnealingLR=object)
    _mod("yacs")
    _mod("yacs.config", load_cfg=lambda *a, **k: None)
    return _AttrDict


def import_reference():
    """Returns (edict, ref_utils, SimCLR, Hybrid2Model)."""
    edict = install_stubs()
    import src.models.utils as ref_utils  # noqa: E402
    from src.models.unsupervised.hybrid2_model import Hybrid2Model  # noqa: E402
    from src.models.unsupervised.simclr_model import SimCLR  # noqa: E402

    return edict, ref_utils, SimCLR, Hybrid2Model


if __name__ == "__main__":
    edict, ru, S, H = import_reference()
    z1 = torch.nn.functional.normalize(torch.randn(4, 128))
    z2 = torch.nn.functional.normalize(torch.randn(4, 128))
    print("reference loss:", float(ru.vanila_contrastive_loss(z1, z2)))
