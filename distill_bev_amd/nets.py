"""Dense (GEMM-shaped) modules of the student / teacher in plain PyTorch-ROCm.

These are MFMA territory served by MIOpen / hipBLASLt through torch (north_star: "MFMA reserved
for the ResNet/SECOND dense convs where it is a real GEMM"); they are NOT hand-written.  Class
names, constructor kwargs and attribute (= checkpoint key) names follow the reference:

  ResNet            mmdet==2.24.0 mmdet/models/backbones/resnet.py (un-vendored; 'pytorch' style,
                    torchvision-compatible keys) -- call site CFG_D:96-107
  FPNForBEVDet      mmdet3d/models/necks/fpn.py:10,60-204
  BasicBlock/Bottleneck  mmdet3d/models/bricks/res_block.py:11-100,102-
  ResNetForBEVDet   mmdet3d/models/backbones/resnet.py:13-62
  FPN_LSS           mmdet3d/models/necks/lss_fpn.py:10-72
  SECOND            mmdet3d/models/backbones/second.py:11-93
  SECONDFPN         mmdet3d/models/necks/second_fpn.py:12-93
  (DCNv2 lives in dcn.py on the gfx950 sampling kernels)
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as cp

from .bn_act import bn_act, bn_act_dual, conv1x1_bn_ready, conv1x1_stats, forked, split_downsample
from .pool import conv_norm_relu_max_pool, max_pool
from .wino import conv3x3_bn_ready, conv3x3_stats
from .registry import (MODELS, ConvModule, build_activation_layer, build_conv_layer, build_norm_layer,
                       build_upsample_layer, register_conv)


def _conv1x1_stats(conv, bn, x):
    """a bottleneck's 1x1 convolution -> (z, partial statistics rows of z or None): the bf16x6 GEMM with the norm's statistics in its
    epilogue where it takes the layer, else the fp32-MFMA GEMM with the same epilogue on the first stage's large maps
    (conv1x1_bn_ready: the round-3 kernel, 0.4 ms per step slower than the bf16x6 one on the layers both take), else the module"""
    from . import bn_act as BA
    from . import gemm_bf6 as G
    if (G._STATS and conv.bias is None and G.eligible(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
            and BA._state["enabled"] and type(bn) in BA._BN_TYPES and bn.affine and bn.training and bn.momentum is not None
            and bn.running_mean is not None and BA._channels_ok(conv.out_channels)):
        return G.conv1x1_stats(x, conv.weight)                # (also without autograd: the detached frame)
    if (G._STATS and conv.bias is None and tuple(conv.stride) == (2, 2)
            and G.eligible_s2(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
            and BA._state["enabled"] and type(bn) in BA._BN_TYPES and bn.affine and bn.training and bn.momentum is not None
            and bn.running_mean is not None and BA._channels_ok(conv.out_channels)):
        return G.conv1x1_s2_stats(x, conv.weight)             # the stride-2 `downsample` convolution: subsample + GEMM + statistics
    if conv1x1_bn_ready(conv, bn, x):
        return conv1x1_stats(conv, x)
    return conv(x), None


def _conv_stats(conv, bn, x):
    """(conv(x), partial statistics rows or None): the Winograd 3x3 kernel's epilogue takes the following norm's batch statistics;
    so does the implicit bf16x6 GEMM of a stride-2 3x3 convolution"""
    if conv3x3_bn_ready(conv, bn, x):
        return conv3x3_stats(x, conv.weight, None)
    from . import bn_act as BA
    from . import gemm_bf6 as G
    if (G._STATS and type(conv) is G.Bf6Conv3x3S2 and conv.bias is None
            and G.eligible_c3s2(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
            and BA._state["enabled"] and type(bn) in BA._BN_TYPES and bn.affine and bn.training and bn.momentum is not None
            and bn.running_mean is not None and BA._channels_ok(conv.out_channels)):
        return G.conv3x3_s2_stats(x, conv.weight)
    return conv(x), None


# --------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style="pytorch",
                 with_cp=False, conv_cfg=None, norm_cfg=dict(type="BN"), dcn=None, plugins=None,
                 init_cfg=None, act_cfg=dict(type="ReLU", inplace=True)):
        super().__init__()
        assert dcn is None and plugins is None
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 3, stride=stride, padding=dilation,
                                      dilation=dilation, bias=False)
        self.add_module(self.norm1_name, norm1)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, padding=1, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.act1 = build_activation_layer(act_cfg)
        self.act2 = build_activation_layer(act_cfg)
        self.downsample = downsample
        self.stride, self.dilation, self.with_cp = stride, dilation, with_cp

    def _inner(self, x):
        identity = x
        out = self.act1(getattr(self, self.norm1_name)(self.conv1(x)))
        out = getattr(self, self.norm2_name)(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return out + identity

    def _fused(self, x):
        """same op sequence with norm -> (+identity) -> relu on the fused kernels (bn_act falls back by itself)"""
        # x feeds the first convolution, its second handle (the previous block's forked output, bn_act.forked) the identity branch:
        # the two gradients reach that block's fused norm backward as two addends instead of being summed by a pass of their own
        xi = forked(x)
        n1, n2 = getattr(self, self.norm1_name), getattr(self, self.norm2_name)
        # 3x3 convolutions on the Winograd kernels hand the norm its batch statistics (no statistics pass over their output)
        z1, p1 = _conv_stats(self.conv1, n1, x)
        out = bn_act(z1, n1, None, True, pre=p1)
        z2, p2 = _conv_stats(self.conv2, n2, out)
        ds = split_downsample(self.downsample)
        if ds is not None:         # norm of the main path and norm of the identity branch, add and ReLU in one pass
            return bn_act_dual(z2, n2, ds[0](xi), ds[1], True, pre=p2, fork=True)
        identity = xi if self.downsample is None else self.downsample(xi)
        return bn_act(z2, n2, identity, True, pre=p2, fork=True)

    def forward(self, x):
        if self.with_cp and x.requires_grad:
            return self.act2(cp.checkpoint(self._inner, x, use_reentrant=False))
        if type(self.act1) is nn.ReLU and type(self.act2) is nn.ReLU:
            return self._fused(x)
        return self.act2(self._inner(x))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style="pytorch",
                 with_cp=False, conv_cfg=None, norm_cfg=dict(type="BN"), dcn=None, plugins=None,
                 init_cfg=None, act_cfg=dict(type="ReLU", inplace=True)):
        super().__init__()
        assert style in ("pytorch", "caffe") and dcn is None and plugins is None
        self.conv1_stride, self.conv2_stride = (1, stride) if style == "pytorch" else (stride, 1)
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.norm3_name, norm3 = build_norm_layer(norm_cfg, planes * self.expansion, postfix=3)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, kernel_size=1, stride=self.conv1_stride, bias=False)
        self.add_module(self.norm1_name, norm1)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, kernel_size=3, stride=self.conv2_stride,
                                      padding=dilation, dilation=dilation, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.conv3 = build_conv_layer(conv_cfg, planes, planes * self.expansion, kernel_size=1, bias=False)
        self.add_module(self.norm3_name, norm3)
        self.act1 = build_activation_layer(act_cfg)
        self.act2 = build_activation_layer(act_cfg)
        self.act3 = build_activation_layer(act_cfg)
        self.downsample = downsample
        self.with_cp = with_cp

    def _inner(self, x):
        identity = x
        out = self.act1(getattr(self, self.norm1_name)(self.conv1(x)))
        out = self.act2(getattr(self, self.norm2_name)(self.conv2(out)))
        out = getattr(self, self.norm3_name)(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return out + identity

    def _fused(self, x):
        """same op sequence with norm -> (+identity) -> relu on the fused kernels (bn_act falls back by itself)"""
        n1, n3 = getattr(self, self.norm1_name), getattr(self, self.norm3_name)
        xi = forked(x)             # second handle of the previous block's output for the identity branch (see BasicBlock._fused)
        # 1x1 convolutions of the large maps: fp32-MFMA GEMM with the norm's batch statistics in its epilogue (no statistics pass)
        z1, p1 = _conv1x1_stats(self.conv1, n1, x)
        out = bn_act(z1, n1, None, True, pre=p1)
        n2 = getattr(self, self.norm2_name)
        z2, p2 = _conv_stats(self.conv2, n2, out)
        out = bn_act(z2, n2, None, True, pre=p2)
        z3, p3 = _conv1x1_stats(self.conv3, n3, out)
        ds = split_downsample(self.downsample)
        if ds is not None:
            zd, pd = _conv1x1_stats(ds[0], ds[1], xi)
            return bn_act_dual(z3, n3, zd, ds[1], True, pre=p3, pre_d=pd, fork=True)
        identity = xi if self.downsample is None else self.downsample(xi)
        return bn_act(z3, n3, identity, True, pre=p3, fork=True)

    def forward(self, x):
        if self.with_cp and x.requires_grad:
            return self.act3(cp.checkpoint(self._inner, x, use_reentrant=False))
        if type(self.act1) is nn.ReLU and type(self.act2) is nn.ReLU and type(self.act3) is nn.ReLU:
            return self._fused(x)
        return self.act3(self._inner(x))


@MODELS.register_module()
class ResNet(nn.Module):
    """mmdet ResNet (depth 18/34/50/101), 'pytorch' style, no deep stem / DCN / plugins."""
    arch_settings = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)),
                     50: (Bottleneck, (3, 4, 6, 3)), 101: (Bottleneck, (3, 4, 23, 3))}

    def __init__(self, depth, in_channels=3, stem_channels=None, base_channels=64, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style="pytorch",
                 deep_stem=False, avg_down=False, frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type="BN", requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False,
                 zero_init_residual=True, pretrained=None, init_cfg=None):
        super().__init__()
        assert not deep_stem and not avg_down and dcn is None and plugins is None
        block, stage_blocks = self.arch_settings[depth]
        stem_channels = stem_channels or base_channels
        self.depth, self.out_indices, self.norm_eval, self.frozen_stages = depth, out_indices, norm_eval, frozen_stages
        self.pretrained, self.init_cfg = pretrained, init_cfg
        self.deep_stem = False
        self.conv1 = build_conv_layer(conv_cfg, in_channels, stem_channels, kernel_size=7, stride=2, padding=3, bias=False)
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, stem_channels, postfix=1)
        self.add_module(self.norm1_name, norm1)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.res_layers = []
        inplanes = stem_channels
        for i, nb in enumerate(stage_blocks[:num_stages]):
            planes = base_channels * 2 ** i
            layers = []
            downsample = None
            if strides[i] != 1 or inplanes != planes * block.expansion:
                downsample = nn.Sequential(
                    build_conv_layer(conv_cfg, inplanes, planes * block.expansion, kernel_size=1,
                                     stride=strides[i], bias=False),
                    build_norm_layer(norm_cfg, planes * block.expansion)[1])
            layers.append(block(inplanes, planes, stride=strides[i], dilation=dilations[i], downsample=downsample,
                                style=style, with_cp=with_cp, conv_cfg=conv_cfg, norm_cfg=norm_cfg))
            inplanes = planes * block.expansion
            for _ in range(1, nb):
                layers.append(block(inplanes, planes, stride=1, dilation=dilations[i], style=style,
                                    with_cp=with_cp, conv_cfg=conv_cfg, norm_cfg=norm_cfg))
            name = f"layer{i + 1}"
            self.add_module(name, nn.Sequential(*layers))
            self.res_layers.append(name)
        self.feat_dim = inplanes
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.ones_(m.weight); nn.init.zeros_(m.bias)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.zeros_(getattr(m, m.norm3_name).weight)
                elif isinstance(m, BasicBlock):
                    nn.init.zeros_(getattr(m, m.norm2_name).weight)
        self._freeze_stages()

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    def forward(self, x):
        x = conv_norm_relu_max_pool(self.conv1, self.norm1, self.maxpool, x)      # conv (+ statistics) -> norm / ReLU / pooling in one pass
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def _freeze_stages(self):
        """mmdet ResNet._freeze_stages: stem and the first `frozen_stages` residual stages in eval mode without grads."""
        if self.frozen_stages >= 0:
            self.norm1.eval()
            for m in (self.conv1, self.norm1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f"layer{i}")
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def init_weights(self):
        """`pretrained` (a path or torchvision:// URL in the reference's configs) / init_cfg(type='Pretrained'): load a
        local state dict when the file exists; otherwise say so -- never silently train from another initialisation."""
        src = self.pretrained
        if src is None and isinstance(self.init_cfg, dict) and self.init_cfg.get("type") == "Pretrained":
            src = self.init_cfg.get("checkpoint")
        if src is None:
            return
        import os
        import warnings
        if isinstance(src, str) and os.path.isfile(src):
            sd = torch.load(src, map_location="cpu")
            sd = sd.get("state_dict", sd)
            missing, unexpected = self.load_state_dict(sd, strict=False)
            missing = [k for k in missing if "num_batches_tracked" not in k]
            if missing:
                raise RuntimeError(f"ResNet pretrained weights {src}: missing keys {missing[:8]}...")
        else:
            warnings.warn(f"ResNet: pretrained weights '{src}' are not reachable from this machine (no network / no such "
                          "file); the backbone keeps its random initialisation", RuntimeWarning)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self


@MODELS.register_module()
class FPNForBEVDet(nn.Module):
    """necks/fpn.py:60-204: laterals for every input level, top-down nearest upsampling, fpn_convs
    only for the levels in out_ids; returns outs[0]."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, out_ids=[],
                 add_extra_convs=False, relu_before_extra_convs=False, no_norm_on_lateral=False,
                 conv_cfg=None, norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode="nearest"), init_cfg=None):
        super().__init__()
        assert not add_extra_convs
        self.in_channels, self.out_channels, self.out_ids = in_channels, out_channels, list(out_ids)
        self.start_level = start_level
        self.backbone_end_level = len(in_channels) if end_level == -1 else end_level
        self.upsample_cfg = dict(upsample_cfg)
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, conv_cfg=conv_cfg,
                                                 norm_cfg=None if no_norm_on_lateral else norm_cfg,
                                                 act_cfg=act_cfg, inplace=False))
            if i in self.out_ids:
                self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, conv_cfg=conv_cfg,
                                                 norm_cfg=norm_cfg, act_cfg=act_cfg, inplace=False))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        laterals = [lc(inputs[i + self.start_level]) for i, lc in enumerate(self.lateral_convs)]
        for i in range(len(laterals) - 1, 0, -1):
            if "scale_factor" in self.upsample_cfg:
                laterals[i - 1] = laterals[i - 1] + F.interpolate(laterals[i], **self.upsample_cfg)
            else:
                laterals[i - 1] = laterals[i - 1] + F.interpolate(laterals[i], size=laterals[i - 1].shape[2:],
                                                                  **self.upsample_cfg)
        outs = [self.fpn_convs[k](laterals[i]) for k, i in enumerate(self.out_ids)]
        return outs[0]


@MODELS.register_module()
class ResNetForBEVDet(nn.Module):
    def __init__(self, numC_input, num_layer=[2, 2, 2], num_channels=None, stride=[2, 2, 2],
                 backbone_output_ids=None, norm_cfg=dict(type="BN"), act_cfg=dict(type="ReLU", inplace=True),
                 with_cp=False, block_type="Basic"):
        super().__init__()
        assert len(num_layer) == len(stride)
        num_channels = [numC_input * 2 ** (i + 1) for i in range(len(num_layer))] if num_channels is None else num_channels
        self.backbone_output_ids = range(len(num_layer)) if backbone_output_ids is None else backbone_output_ids
        layers = []
        cur = numC_input
        for i in range(len(num_layer)):
            if block_type == "BottleNeck":
                layer = [Bottleneck(cur, num_channels[i] // 4, stride=stride[i],
                                    downsample=nn.Conv2d(cur, num_channels[i], 3, stride[i], 1),
                                    norm_cfg=norm_cfg, act_cfg=act_cfg)]
                cur = num_channels[i]
                layer += [Bottleneck(cur, cur // 4, norm_cfg=norm_cfg, act_cfg=act_cfg) for _ in range(num_layer[i] - 1)]
            else:
                assert block_type == "Basic"
                layer = [BasicBlock(cur, num_channels[i], stride=stride[i],
                                    downsample=nn.Conv2d(cur, num_channels[i], 3, stride[i], 1),
                                    norm_cfg=norm_cfg, act_cfg=act_cfg)]
                cur = num_channels[i]
                layer += [BasicBlock(cur, cur, norm_cfg=norm_cfg, act_cfg=act_cfg) for _ in range(num_layer[i] - 1)]
            layers.append(nn.Sequential(*layer))
        self.layers = nn.Sequential(*layers)
        self.with_cp = with_cp

    def forward(self, x):
        feats = []
        for lid, layer in enumerate(self.layers):
            x = cp.checkpoint(layer, x, use_reentrant=False) if self.with_cp else layer(x)
            if lid in self.backbone_output_ids:
                feats.append(x)
        return feats


@MODELS.register_module()
class FPN_LSS(nn.Module):
    def __init__(self, in_channels, out_channels, scale_factor=4, input_feature_index=(0, 2),
                 norm_cfg=dict(type="BN"), extra_upsample=2, lateral=None, extra_norm_act=False,
                 act_cfg=dict(type="ReLU", inplace=True)):
        super().__init__()
        self.input_feature_index = input_feature_index
        self.extra_upsample = extra_upsample is not None
        self.up = nn.Upsample(scale_factor=scale_factor, mode="bilinear", align_corners=True)
        cf = 2 if self.extra_upsample else 1
        self.conv = nn.Sequential(
            nn.Conv2d(in_channels, out_channels * cf, kernel_size=3, padding=1, bias=False),
            build_norm_layer(norm_cfg, out_channels * cf, postfix=0)[1], build_activation_layer(act_cfg),
            nn.Conv2d(out_channels * cf, out_channels * cf, kernel_size=3, padding=1, bias=False),
            build_norm_layer(norm_cfg, out_channels * cf, postfix=0)[1], build_activation_layer(act_cfg))
        if self.extra_upsample:
            up2 = [nn.Upsample(scale_factor=extra_upsample, mode="bilinear", align_corners=True),
                   nn.Conv2d(out_channels * cf, out_channels, kernel_size=3, padding=1, bias=False),
                   build_norm_layer(norm_cfg, out_channels, postfix=0)[1], build_activation_layer(act_cfg),
                   nn.Conv2d(out_channels, out_channels, kernel_size=1, padding=0)]
            if extra_norm_act:
                up2 += [build_norm_layer(norm_cfg, out_channels, postfix=0)[1], build_activation_layer(act_cfg)]
            self.up2 = nn.Sequential(*up2)
        self.lateral = lateral is not None
        if self.lateral:
            self.lateral_conv = nn.Sequential(nn.Conv2d(lateral, lateral, kernel_size=1, padding=0, bias=False),
                                              build_norm_layer(norm_cfg, lateral, postfix=0)[1],
                                              build_activation_layer(act_cfg))

    def forward(self, feats):
        x2, x1 = feats[self.input_feature_index[0]], feats[self.input_feature_index[1]]
        if self.lateral:
            x2 = self.lateral_conv(x2)
        x = self.conv(torch.cat([x2, self.up(x1)], dim=1))
        if self.extra_upsample:
            x = self.up2(x)
        return x


@MODELS.register_module()
class SECOND(nn.Module):
    def __init__(self, in_channels=128, out_channels=[128, 128, 256], layer_nums=[3, 5, 5], layer_strides=[2, 2, 2],
                 norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), conv_cfg=dict(type="Conv2d", bias=False),
                 init_cfg=None, pretrained=None, act_cfg=dict(type="ReLU", inplace=True)):
        super().__init__()
        assert len(layer_strides) == len(layer_nums) == len(out_channels)
        in_filters = [in_channels, *out_channels[:-1]]
        blocks = []
        for i, layer_num in enumerate(layer_nums):
            block = [build_conv_layer(conv_cfg, in_filters[i], out_channels[i], 3, stride=layer_strides[i], padding=1),
                     build_norm_layer(norm_cfg, out_channels[i])[1], build_activation_layer(act_cfg)]
            for _ in range(layer_num):
                block += [build_conv_layer(conv_cfg, out_channels[i], out_channels[i], 3, padding=1),
                          build_norm_layer(norm_cfg, out_channels[i])[1], build_activation_layer(act_cfg)]
            blocks.append(nn.Sequential(*block))
        self.blocks = nn.ModuleList(blocks)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        outs = []
        for blk in self.blocks:
            x = blk(x)
            outs.append(x)
        return tuple(outs)


@MODELS.register_module()
class SECONDFPN(nn.Module):
    def __init__(self, in_channels=[128, 128, 256], out_channels=[256, 256, 256], upsample_strides=[1, 2, 4],
                 norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), upsample_cfg=dict(type="deconv", bias=False),
                 conv_cfg=dict(type="Conv2d", bias=False), use_conv_for_no_stride=False, init_cfg=None,
                 act_cfg=dict(type="ReLU", inplace=True)):
        super().__init__()
        assert len(out_channels) == len(upsample_strides) == len(in_channels)
        self.in_channels, self.out_channels = in_channels, out_channels
        deblocks = []
        for i, oc in enumerate(out_channels):
            stride = upsample_strides[i]
            if stride > 1 or (stride == 1 and not use_conv_for_no_stride):
                up = build_upsample_layer(upsample_cfg, in_channels=in_channels[i], out_channels=oc,
                                          kernel_size=upsample_strides[i], stride=upsample_strides[i])
            else:
                s = int(np.round(1 / stride).astype(np.int64))
                up = build_conv_layer(conv_cfg, in_channels=in_channels[i], out_channels=oc, kernel_size=s, stride=s)
            deblocks.append(nn.Sequential(up, build_norm_layer(norm_cfg, oc)[1], build_activation_layer(act_cfg)))
        self.deblocks = nn.ModuleList(deblocks)
        for m in self.modules():
            if isinstance(m, (nn.ConvTranspose2d, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        assert len(x) == len(self.in_channels)
        ups = [d(x[i]) for i, d in enumerate(self.deblocks)]
        return [torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]]


class SELikeModule(nn.Module):
    """view_transformer_mine.py:267-280."""

    def __init__(self, in_channel=512, feat_channel=256, intrinsic_channel=33):
        super().__init__()
        self.input_conv = nn.Conv2d(in_channel, feat_channel, kernel_size=1, padding=0)
        self.fc = nn.Sequential(nn.BatchNorm1d(intrinsic_channel), nn.Linear(intrinsic_channel, feat_channel), nn.Sigmoid())

    def forward(self, x, cam_params):
        x = self.input_conv(x)
        b, c = x.shape[:2]
        return x * self.fc(cam_params).view(b, c, 1, 1)
