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

    def _extras(cfg):
        return {kk: vv for kk, vv in (cfg or {}).items() if kk != "type"}

    def build_conv_layer(cfg, *a, **k):          # mmcv: layer(*args, **kwargs, **cfg_without_type)
        return nn.Conv2d(*a, **k, **_extras(cfg))

    def build_norm_layer(cfg, n, postfix=""):
        t = cfg.get("type", "BN")
        kw = {k: v for k, v in cfg.items() if k in ("eps", "momentum")}
        cls = nn.BatchNorm1d if t == "BN1d" else nn.BatchNorm2d
        return "bn", cls(n, **kw)

    def build_upsample_layer(cfg, *a, **k):
        return nn.ConvTranspose2d(*a, **k, **_extras(cfg))

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


# =====================================================================================
# Round 2: stubs wide enough to import the detector / head / voxel-encoder files by path
# (bevdet_distill.py, bevdet_distill_more.py, centerpoint_head.py, pillar_encoder.py,
#  second.py, second_fpn.py, ops/voxel/scatter_points.py).  Everything a fixture records is
# computed by the reference's own python; the stubs below provide
#   * names the files import at module level and never touch on the functions we call
#     (cv2, PIL, mmcv.Config, load_checkpoint, the sibling detectors, nms, bbox coders ...);
#   * the un-vendored third-party arithmetic the reference is CONFIGURED with (mmdet==2.24.0
#     MSELoss / L1Loss / GaussianFocalLoss / multi_apply; mmcv ConvModule) restated from
#     their published definitions -- named as such in DESIGN.md;
#   * the CUDA-only extension entry points of ops/voxel (dynamic_point_to_voxel_forward /
#     backward) written with the very ATen calls of the reference's host code
#     (scatter_points_cuda.cu:183-239: masked_fill -> unique_dim(sorted) -> drop the (-1)
#     row -> reduce), i.e. torch.unique; independent of oracle/voxel.c.
# =====================================================================================
import functools  # noqa: E402

import numpy as np  # noqa: E402
import torch.nn.functional as F  # noqa: E402


# ---- mmdet 2.24.0 losses (mmdet/models/losses/{utils,mse_loss,smooth_l1_loss,gaussian_focal_loss}.py) ----
def _weight_reduce_loss(loss, weight=None, reduction="mean", avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == "mean":
            return loss.mean()
        if reduction == "sum":
            return loss.sum()
        return loss
    if reduction == "mean":
        eps = torch.finfo(torch.float32).eps
        return loss.sum() / (avg_factor + eps)
    if reduction != "none":
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


class _MmdetLoss(nn.Module):
    def __init__(self, reduction="mean", loss_weight=1.0, **kw):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight
        self.__dict__.update(kw)

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        red = reduction_override if reduction_override else self.reduction
        return self.loss_weight * _weight_reduce_loss(self.elementwise(pred, target), weight, red, avg_factor)


class MSELoss(_MmdetLoss):
    def elementwise(self, pred, target):
        return F.mse_loss(pred, target, reduction="none")


class L1Loss(_MmdetLoss):
    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        if target.numel() == 0:
            return pred.sum() * 0
        return super().forward(pred, target, weight, avg_factor, reduction_override)

    def elementwise(self, pred, target):
        return torch.abs(pred - target)


class GaussianFocalLoss(_MmdetLoss):
    def __init__(self, alpha=2.0, gamma=4.0, reduction="mean", loss_weight=1.0):
        super().__init__(reduction, loss_weight)
        self.alpha, self.gamma = alpha, gamma

    def elementwise(self, pred, target):
        eps = 1e-12
        pos_weights = target.eq(1)
        neg_weights = (1 - target).pow(self.gamma)
        pos_loss = -(pred + eps).log() * (1 - pred).pow(self.alpha) * pos_weights
        neg_loss = -(1 - pred + eps).log() * pred.pow(self.alpha) * neg_weights
        return pos_loss + neg_loss


_LOSSES = dict(MSELoss=MSELoss, L1Loss=L1Loss, GaussianFocalLoss=GaussianFocalLoss)


def build_loss(cfg):
    cfg = dict(cfg)
    return _LOSSES[cfg.pop("type")](**cfg)


def multi_apply(func, *args, **kwargs):
    """mmdet/core/utils/misc.py"""
    pfunc = functools.partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


class LiDARBoxesStub:
    """Members of LiDARInstance3DBoxes the hot path reads (lidar_box3d.py:41-47): .tensor [M,9] bottom-centre boxes and
    .gravity_center.  (The class itself imports the iou3d .so at module level -> not importable.)"""

    def __init__(self, tensor):
        self.tensor = torch.as_tensor(tensor, dtype=torch.float32)

    @property
    def gravity_center(self):
        bottom_center = self.tensor[:, :3]
        gravity_center = torch.zeros_like(bottom_center)
        gravity_center[:, :2] = bottom_center[:, :2]
        gravity_center[:, 2] = bottom_center[:, 2] + self.tensor[:, 5] * 0.5
        return gravity_center


class LiDARPointsStub:
    """core/points/base_points.py: .coord = tensor[:, :3]"""

    def __init__(self, tensor, points_dim=3, attribute_dims=None):
        self.tensor = torch.as_tensor(tensor)

    @property
    def coord(self):
        return self.tensor[:, :3]


def _voxel_layer_stub():
    """dynamic_point_to_voxel_forward/backward with the reference host code's own ATen calls
    (scatter_points_cuda.cu:183-239, :241-308)."""

    def dynamic_point_to_voxel_forward(feats, coors, reduce_type):
        if feats.size(0) == 0:
            return [feats.clone().detach(), coors.clone().detach(), coors.new_empty((0,), dtype=torch.int32),
                    coors.new_empty((0,), dtype=torch.int32)]
        coors_clean = coors.masked_fill(coors.lt(0).any(-1, True), -1)
        out_coors, coors_map, reduce_count = torch.unique(coors_clean, dim=0, sorted=True, return_inverse=True,
                                                          return_counts=True)
        if bool(out_coors[0, 0].lt(0)):
            out_coors = out_coors[1:]
            reduce_count = reduce_count[1:]
            coors_map = coors_map - 1
        coors_map = coors_map.to(torch.int32)
        reduce_count = reduce_count.to(torch.int32)
        M, C = out_coors.size(0), feats.size(1)
        valid = coors_map >= 0
        idx = coors_map[valid].long()
        if reduce_type == "max":
            red = feats.new_full((M, C), float("-inf"))
            red.index_reduce_(0, idx, feats[valid], "amax", include_self=True)
        else:
            # feats_reduce_kernel adds with float atomics (order undefined, ~5e-7 per the reference's own doc,
            # scatter_points.py:60-61); the fixture sums in fp64 and rounds once = the value every order is near
            red = feats.new_zeros((M, C), dtype=torch.float64)
            red.index_add_(0, idx, feats[valid].double())
            if reduce_type == "mean":
                red = red / reduce_count.unsqueeze(-1).double()
            red = red.to(feats.dtype)
        return [red, out_coors, coors_map, reduce_count]

    def dynamic_point_to_voxel_backward(grad_feats, grad_reduced, feats, reduced, coors_map, reduce_count, reduce_type):
        N = feats.size(0)
        if N == 0 or reduced.size(0) == 0:
            return
        valid = coors_map >= 0
        idx = coors_map.long().clamp(min=0)
        if reduce_type in ("mean", "sum"):
            g = grad_reduced[idx]
            if reduce_type == "mean":
                g = g / reduce_count[idx].unsqueeze(-1).to(g.dtype)
            grad_feats.copy_(torch.where(valid.unsqueeze(-1), g, torch.zeros_like(g)))
        else:
            # argmax traceback: lowest point index among the points equal to the max (atomicMin, :135-160)
            hit = (feats == reduced[idx]) & valid.unsqueeze(-1)
            pid = torch.arange(N).unsqueeze(-1).expand_as(hit)
            big = torch.full((reduced.size(0), feats.size(1)), N, dtype=torch.long)
            cand = torch.where(hit, pid, torch.full_like(pid, N))
            big.scatter_reduce_(0, idx.unsqueeze(-1).expand_as(cand), cand, "amin", include_self=True)
            win = hit & (pid == big[idx])
            grad_feats.copy_(torch.where(win, grad_reduced[idx], torch.zeros_like(feats)))

    return dict(dynamic_point_to_voxel_forward=dynamic_point_to_voxel_forward,
                dynamic_point_to_voxel_backward=dynamic_point_to_voxel_backward)


def install_full_stubs():
    install_stubs()
    if getattr(sys.modules["mmcv"], "_dbev_full", False):
        return
    mmcv = sys.modules["mmcv"]
    mmcv._dbev_full = True
    mmcv.Config = type("Config", (), {"fromfile": staticmethod(lambda p: (_ for _ in ()).throw(RuntimeError("stub")))})

    class ConvModule(nn.Module):
        """mmcv/cnn/bricks/conv_module.py for order ('conv','norm','act')"""

        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                     bias="auto", conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), inplace=True, **kw):
            super().__init__()
            with_norm = norm_cfg is not None
            if bias == "auto":
                bias = not with_norm
            self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
            self.bn = sys.modules["mmcv.cnn"].build_norm_layer(norm_cfg, out_channels)[1] if with_norm else None
            self.activate = nn.ReLU(inplace=inplace) if act_cfg is not None else None

        def forward(self, x):
            x = self.conv(x)
            if self.bn is not None:
                x = self.bn(x)
            return self.activate(x) if self.activate is not None else x

    mmcv.cnn.ConvModule = ConvModule
    mmcv.runner.load_checkpoint = lambda *a, **k: None
    mmcv.runner.load_state_dict = lambda *a, **k: None

    reg = _Registry()
    _mod("cv2")
    pil = _mod("PIL"); pil.Image = _mod("PIL.Image")
    mmdet = _mod("mmdet"); mmdet.__path__ = []
    _mod("mmdet.models", DETECTORS=reg, BACKBONES=reg, NECKS=reg, HEADS=reg, build_detector=lambda cfg: None,
         ResNet=type("ResNet", (nn.Module,), {}))
    _mod("mmdet.utils", get_root_logger=lambda *a, **k: None)
    _mod("mmdet.core", multi_apply=multi_apply, build_bbox_coder=lambda cfg: None)

    # real reference files for the small pure-python helpers
    G = gaussian()
    cs = load("mmdet3d/models/utils/clip_sigmoid.py", "refpkg_clip_sigmoid")
    bnp = box_np_ops()
    m3 = _mod("mmdet3d"); m3.__path__ = []
    core = _mod("mmdet3d.core", circle_nms=None, xywhr2xyxyr=None, **{k: getattr(G, k) for k in (
        "draw_heatmap_gaussian", "gaussian_radius", "centerpoint_radius_func1", "centerpoint_radius_func2",
        "centerpoint_radius_func3", "maxwh_radius_func")})
    core.__path__ = []
    _mod("mmdet3d.core.points", LiDARPoints=LiDARPointsStub)
    _mod("mmdet3d.core.bbox", LiDARInstance3DBoxes=LiDARBoxesStub, box_np_ops=bnp)
    mm = _mod("mmdet3d.models"); mm.__path__ = []
    mbuilder = _mod("mmdet3d.models.builder", HEADS=reg, build_loss=build_loss, build_head=lambda cfg: None)
    mm.builder = mbuilder
    _mod("mmdet3d.models.utils", clip_sigmoid=cs.clip_sigmoid)
    ops = _mod("mmdet3d.ops"); ops.__path__ = []
    iou = _mod("mmdet3d.ops.iou3d"); iou.__path__ = []
    _mod("mmdet3d.ops.iou3d.iou3d_utils", nms_gpu=None)

    # ops/voxel/scatter_points.py is loaded for real on top of the ext stub
    vpk = _mod("refpkg.ops"); vpk.__path__ = []
    vox = _mod("refpkg.ops.voxel"); vox.__path__ = []
    _mod("refpkg.ops.voxel.voxel_layer", **_voxel_layer_stub())
    sp = load("mmdet3d/ops/voxel/scatter_points.py", "refpkg.ops.voxel.scatter_points")
    ops.DynamicScatter = sp.DynamicScatter
    ops.dynamic_scatter = sp.dynamic_scatter

    # sibling modules the detector files import from
    b = sys.modules["refpkg.models.builder"]
    b.build_neck = lambda cfg: None
    b.build_loss = build_loss
    det = _mod("refpkg.models.detectors"); det.__path__ = []
    base = type("BaseDet", (nn.Module,), {})
    _mod("refpkg.models.detectors.centerpoint", CenterPoint=type("CenterPoint", (base,), {}))
    _mod("refpkg.models.detectors.bevdet", BEVDet=type("BEVDet", (base,), {}),
         BEVDepth4D=type("BEVDepth4D", (base,), {}), BEVDetSequentialES=type("BEVDetSequentialES", (base,), {}))
    _mod("refpkg.models.bricks", NonLocalBlockND=type("NonLocalBlockND", (nn.Module,), {}))
    sys.modules["refpkg.models.necks"].ViewTransformerLSSBEVDepthReproduce = None
    sys.modules["refpkg.models"].builder = b
    dh = _mod("refpkg.models.dense_heads"); dh.__path__ = []


def scatter_points():
    install_full_stubs()
    return sys.modules["refpkg.ops.voxel.scatter_points"]


def bevdet_distill():
    install_full_stubs()
    return load("mmdet3d/models/detectors/bevdet_distill.py", "refpkg.models.detectors.bevdet_distill")


def bevdet_distill_more():
    bevdet_distill()
    return load("mmdet3d/models/detectors/bevdet_distill_more.py", "refpkg.models.detectors.bevdet_distill_more")


def centerpoint_head():
    install_full_stubs()
    return load("mmdet3d/models/dense_heads/centerpoint_head.py", "refpkg.models.dense_heads.centerpoint_head")


def pillar_encoder():
    install_full_stubs()
    load("mmdet3d/models/voxel_encoders/utils.py", "refpkg.models.voxel_encoders.utils")
    return load("mmdet3d/models/voxel_encoders/pillar_encoder.py", "refpkg.models.voxel_encoders.pillar_encoder")


def second():
    install_full_stubs()
    return load("mmdet3d/models/backbones/second.py", "refpkg.models.backbones.second")


def second_fpn():
    install_full_stubs()
    return load("mmdet3d/models/necks/second_fpn.py", "refpkg.models.necks.second_fpn")


def bare(cls, **attrs):
    """An instance of a reference class without running its __init__ (which would build the whole mmdet model zoo):
    nn.Module state only, then the attributes the called methods read."""
    obj = cls.__new__(cls)
    nn.Module.__init__(obj)
    for k, v in attrs.items():
        object.__setattr__(obj, k, v) if not isinstance(v, nn.Module) else nn.Module.__setattr__(obj, k, v)
    return obj
