"""Minimal registry + layer builders: the config surface of the reference
(``mmdet3d/models/builder.py:6-99`` re-exporting mmcv/mmdet registries, mmcv.cnn build_*_layer,
mmcv.cnn.ConvModule) without mmcv.  Type names and kwargs are the reference's, so the model
dicts of ``configs/lidar2camera_bev_distillation/*`` build unchanged.
"""
import copy

import torch.nn as nn


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f"{key} already registered in {self.name}")
            self._modules[key] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def get(self, key):
        return self._modules.get(key)

    def __contains__(self, key):
        return key in self._modules

    def build(self, cfg, **default_args):
        if cfg is None:
            return None
        if isinstance(cfg, nn.Module):
            return cfg
        cfg = copy.deepcopy(dict(cfg))
        for k, v in default_args.items():
            cfg.setdefault(k, v)
        typ = cfg.pop("type")
        cls = typ if isinstance(typ, type) else self._modules.get(typ)
        if cls is None:
            raise KeyError(f"'{typ}' is not registered in the {self.name} registry "
                           f"(known: {sorted(self._modules)})")
        return cls(**cfg)


# mmdet3d/models/builder.py: BACKBONES/NECKS/... are shared mmdet registries; one MODELS table
# with role aliases gives the same lookup behaviour.
MODELS = Registry("models")
BACKBONES = NECKS = HEADS = DETECTORS = LOSSES = VOXEL_ENCODERS = MIDDLE_ENCODERS = FUSION_LAYERS = MODELS
BBOX_CODERS = Registry("bbox_coder")


def build_backbone(cfg):
    return MODELS.build(cfg)


build_neck = build_head = build_loss = build_voxel_encoder = build_middle_encoder = build_backbone


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """builder.py:53-66."""
    extra = {}
    if train_cfg is not None:
        extra["train_cfg"] = train_cfg
    if test_cfg is not None:
        extra["test_cfg"] = test_cfg
    return MODELS.build(cfg, **extra)


build_model = build_detector


# ---- mmcv.cnn layer builders -------------------------------------------------------------
_NORMS = {"BN": nn.BatchNorm2d, "BN1d": nn.BatchNorm1d, "BN2d": nn.BatchNorm2d, "BN3d": nn.BatchNorm3d,
          "SyncBN": nn.BatchNorm2d, "GN": nn.GroupNorm, "LN": nn.LayerNorm}
_NORM_ABBR = {"BN": "bn", "BN1d": "bn", "BN2d": "bn", "BN3d": "bn", "SyncBN": "bn", "GN": "gn", "LN": "ln"}
_CONVS = {}


def register_conv(name, cls):
    _CONVS[name] = cls


def build_conv_layer(cfg, *args, **kwargs):
    cfg = dict(cfg) if cfg is not None else dict(type="Conv2d")
    typ = cfg.pop("type", "Conv2d")
    if typ in ("Conv2d", "Conv"):
        return nn.Conv2d(*args, **kwargs, **cfg)
    if typ == "Conv1d":
        return nn.Conv1d(*args, **kwargs, **cfg)
    if typ in _CONVS:
        return _CONVS[typ](*args, **kwargs, **cfg)
    raise KeyError(f"unknown conv layer type {typ}")


def build_norm_layer(cfg, num_features, postfix=""):
    cfg = dict(cfg)
    typ = cfg.pop("type")
    requires_grad = cfg.pop("requires_grad", True)
    cfg.setdefault("eps", 1e-5)
    if typ == "GN":
        layer = nn.GroupNorm(num_channels=num_features, **cfg)
    elif typ == "SyncBN":
        # mmcv maps 'SyncBN' to torch.nn.SyncBatchNorm.  In a process group it keeps that meaning (statistics
        # all-reduced over the ranks); a single process has nothing to synchronise with and gets the local
        # BatchNorm2d (same parameters, buffers and state-dict keys), which the fused norm-act kernels cover.
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            layer = nn.SyncBatchNorm(num_features, **cfg)
        else:
            layer = nn.BatchNorm2d(num_features, **cfg)
    else:
        layer = _NORMS[typ](num_features, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return _NORM_ABBR[typ] + str(postfix), layer


def build_upsample_layer(cfg, *args, **kwargs):
    cfg = dict(cfg)
    typ = cfg.pop("type")
    if typ == "deconv":
        return nn.ConvTranspose2d(*args, **kwargs, **cfg)
    if typ in ("nearest", "bilinear"):
        return nn.Upsample(*args, mode=typ, **kwargs, **cfg)
    raise KeyError(f"unknown upsample layer type {typ}")


def build_activation_layer(cfg):
    cfg = dict(cfg)
    typ = cfg.pop("type")
    table = {"ReLU": nn.ReLU, "LeakyReLU": nn.LeakyReLU, "GELU": nn.GELU, "Sigmoid": nn.Sigmoid,
             "SiLU": nn.SiLU, "Tanh": nn.Tanh}
    if typ not in ("ReLU", "LeakyReLU", "SiLU"):
        cfg.pop("inplace", None)
    return table[typ](**cfg)


class ConvModule(nn.Module):
    """mmcv.cnn.ConvModule (order conv -> norm -> act); attribute names conv / bn / activate so
    that checkpoint keys match."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias="auto", conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), inplace=True):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == "auto":
            bias = not self.with_norm
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                     padding=padding, dilation=dilation, groups=groups, bias=bias)
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            act_cfg = dict(act_cfg)
            if act_cfg["type"] in ("ReLU", "LeakyReLU"):
                act_cfg.setdefault("inplace", inplace)
            self.activate = build_activation_layer(act_cfg)
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode="fan_out", nonlinearity="relu")
        if getattr(self.conv, "bias", None) is not None:
            nn.init.zeros_(self.conv.bias)

    @property
    def norm(self):
        return getattr(self, self.norm_name) if self.with_norm else None

    def forward(self, x):
        if self.with_norm:
            # a 3x3 convolution on the Winograd kernels hands the fused norm(+ReLU) behind it the batch statistics of its output
            # (no statistics pass); decided per call from the modules' types / modes and the tensor (wino.conv3x3_bn_ready)
            from .bn_act import BatchNormAct2d, bn_act
            from .wino import conv3x3_bn_ready, conv3x3_stats, conv_norm_relu_eval, fold_ready
            if fold_ready(self.conv, self.norm, x):          # frozen stack: convolution, eval-mode norm and ReLU in one launch
                x = conv_norm_relu_eval(x, self.conv, self.norm)
                return self.activate(x) if self.with_activation else x
            if type(self.norm) is BatchNormAct2d and conv3x3_bn_ready(self.conv, self.norm, x):
                z, rows = conv3x3_stats(x, self.conv.weight, None)
                x = bn_act(z, self.norm, None, True, pre=rows)
                return self.activate(x) if self.with_activation else x
        x = self.conv(x)
        if self.with_norm:
            x = self.norm(x)
        if self.with_activation:
            x = self.activate(x)
        return x
