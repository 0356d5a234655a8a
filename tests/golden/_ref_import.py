"""Import hot-path files of the reference BY PATH with third-party stubs.

Used only by tests/golden/make_golden.py in the build container (where
/root/reference exists).  Nothing here is product code and nothing here runs on
the GPU box.  The stubs only give the reference's *own* code the names it
imports from mmcv / mmdet / torch_scatter / numba (none of which is installed);
all arithmetic that ends up in a fixture is executed by the reference's files.
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get("DBEV_REFERENCE", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Registry:
    def register_module(self, *a, **k):
        def deco(cls):
            return cls
        return deco

    def build(self, cfg, *a, **k):
        raise RuntimeError("registry stub: build() not available in the golden generator")


def _identity_decorator_factory(*a, **k):
    def deco(f):
        return f
    return deco


def install_stubs():
    if "mmcv" in sys.modules and getattr(sys.modules["mmcv"], "_dbev_stub", False):
        return

    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()

    def build_conv_layer(cfg, *a, **k):
        return nn.Conv2d(*a, **k)

    def build_norm_layer(cfg, n, postfix=""):
        t = cfg.get("type", "BN")
        kw = {k: v for k, v in cfg.items() if k in ("eps", "momentum")}
        cls = nn.BatchNorm1d if t == "BN1d" else nn.BatchNorm2d
        return "bn", cls(n, **kw)

    def build_upsample_layer(cfg, *a, **k):
        return nn.ConvTranspose2d(*a, **k)

    def build_activation_layer(cfg):
        return nn.ReLU(inplace=cfg.get("inplace", False))

    mmcv = _mod("mmcv", _dbev_stub=True)
    mmcv.runner = _mod("mmcv.runner", BaseModule=BaseModule,
                       force_fp32=_identity_decorator_factory,
                       auto_fp16=_identity_decorator_factory)
    mmcv.cnn = _mod("mmcv.cnn", build_conv_layer=build_conv_layer,
                    build_norm_layer=build_norm_layer,
                    build_upsample_layer=build_upsample_layer,
                    build_activation_layer=build_activation_layer)

    def scatter_sum(src, index, dim=0):
        out = src.new_zeros((int(index.max()) + 1,) + tuple(src.shape[1:]))
        return out.index_add_(0, index, src)

    _mod("torch_scatter", scatter_sum=scatter_sum)

    def _jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    _mod("numba", jit=_jit, njit=_jit)

    # fake package shell so that relative imports resolve
    pkg = _mod("refpkg")
    pkg.__path__ = []
    models = _mod("refpkg.models")
    models.__path__ = []
    reg = _Registry()
    builder = _mod("refpkg.models.builder", NECKS=reg, BACKBONES=reg, MIDDLE_ENCODERS=reg,
                   VOXEL_ENCODERS=reg, build_backbone=lambda cfg: nn.Identity())
    models.builder = builder
    for sub in ("necks", "middle_encoders", "backbones", "voxel_encoders"):
        m = _mod("refpkg.models." + sub)
        m.__path__ = []


def load(rel_path, modname):
    """Load /root/reference/<rel_path> as module <modname> (e.g.
    'refpkg.models.necks.view_transformer_mine')."""
    install_stubs()
    path = os.path.join(REF, rel_path)
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def vt_mine():
    return load("mmdet3d/models/necks/view_transformer_mine.py",
                "refpkg.models.necks.view_transformer_mine")


def box_np_ops():
    return load("mmdet3d/core/bbox/box_np_ops.py", "refpkg_box_np_ops")


def gaussian():
    return load("mmdet3d/core/utils/gaussian.py", "refpkg_gaussian")


def pillar_scatter():
    return load("mmdet3d/models/middle_encoders/pillar_scatter.py",
                "refpkg.models.middle_encoders.pillar_scatter")
