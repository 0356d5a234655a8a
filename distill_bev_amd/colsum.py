"""Bias terms of the library convolutions (csrc/colsum.hip).

Host-side mirror of what the reference reaches through ``nn.Conv2d(bias=True)`` -> cuDNN in the necks, the view transformer, the BEV
encoder and the adaptation layers (mmdet3d/models/necks/fpn.py:77-95, necks/view_transformer_mine.py:288-309, backbones/resnet.py:80-96,
detectors/bevdet_distill.py:99-132).  ATen gives a MIOpen convolution its bias in a separate pass and sums the bias gradient with a
generic reduction (1.5-2.5 TB/s on channels-last tensors, 0.05 TB/s for the 59- and 27-channel maps of the depth head):

* ``channel_sum(t)``: ``t.sum((0, 2, 3))`` of a channels-last fp32 device tensor as one streaming pass with a fixed summation order;
* ``BiasSumConv2d``: nn.Conv2d whose bias gradient is that pass (forward, data and weight gradient: the library's kernels, as before);
* ``conv_bn_cancelled_bias``: a convolution with a bias followed by a TRAINING-mode BatchNorm -- the norm subtracts the batch mean, so
  the bias changes neither the output nor any other gradient; the convolution runs without it, the running mean gets
  ``momentum * bias`` (it tracks mean(conv) + bias), and the bias gradient, a sum of the norm's input gradient that is zero by
  construction (the reference computes its rounding noise), is returned as zeros."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib as L

_ON = os.environ.get("DBEV_BIAS_SUM", "1") != "0"          # 0: ATen's reduction / bias passes everywhere (A/B runs)


def _nhwc(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def eligible(t):
    return torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.numel() > 0 and (
        _nhwc(t) or (t.dim() == 2 and t.is_contiguous()))


def channel_sum(t):
    """t [N, C, H, W] channels-last (or [M, C] contiguous), fp32, device -> f32[C] = t.sum((0, 2, 3)); not differentiable"""
    dev = L.require_cuda(t)
    if not _ON:
        return t.sum((0, 2, 3)) if t.dim() == 4 else t.sum(0)
    if not eligible(t):
        raise L.DbevHipError("channel_sum: a channels-last [N, C, H, W] (or contiguous [M, C]) fp32 device tensor is required")
    C = int(t.shape[1])
    M = t.numel() // C
    nbytes = int(L.call("dbev_channel_sum_workspace_bytes", M, C))
    if nbytes == 0:
        raise L.DbevHipError(f"channel_sum: unsupported size {M} x {C}")
    out = torch.empty((C,), dtype=torch.float32, device=dev)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.call("dbev_channel_sum_nhwc", L.ptr(t), M, C, L.ptr(out), L.ptr(ws), nbytes, L.stream_ptr(dev))
    return out


class _BiasConv(Function):
    """F.conv2d(x, weight, bias, ...) on the library's kernels; backward: the library's data / weight gradients + channel_sum"""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, groups):
        y = F.conv2d(x, weight, None, stride, padding, dilation, groups)
        y.add_(bias.view(1, -1, 1, 1))                       # what ATen does behind miopen_convolution
        ctx.save_for_backward(x, weight)
        ctx.geom = (list(stride), list(padding), list(dilation), groups)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, padding, dilation, groups = ctx.geom
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = gw = gb = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gx, gw, _ = torch.ops.aten.convolution_backward(gy, x, weight, None, stride, padding, dilation, False, [0, 0], groups,
                                                            [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
        if ctx.needs_input_grad[2]:
            gb = channel_sum(gy)
        return gx, gw, gb, None, None, None, None


class BiasSumConv2d(nn.Conv2d):
    """nn.Conv2d with a bias whose bias gradient is `channel_sum` of the output gradient when the input is a channels-last fp32 device
    tensor and gradients are recorded; the stock module otherwise.  Same parameters and state-dict keys."""

    def forward(self, x):
        if (self.bias is not None and self.padding_mode == "zeros" and torch.is_tensor(x) and x.is_cuda
                and x.dtype == torch.float32 and _nhwc(x) and x.numel() > 0 and not isinstance(self.padding, str)):
            from . import gemm_bf6 as G                       # a 1x1 layer the bf16x6 GEMM takes: its kernels + this module's bias handling
            grad = torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad or self.bias.requires_grad)
            if G.eligible(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
                if grad:
                    return G.conv1x1(x, self.weight, self.bias)
                return G.product(x, self.weight).add_(self.bias.view(1, -1, 1, 1))
            if grad:
                return _BiasConv.apply(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        return super().forward(x)


def use_bias_sum_convs(model):
    """Re-class the plain nn.Conv2d modules that carry a bias (run AFTER the Winograd / skinny / GEMM re-classing: those keep their own
    bias handling); returns how many.  Idempotent."""
    n = 0
    if not _ON:
        return 0
    for m in model.modules():
        if type(m) is nn.Conv2d and m.bias is not None and m.padding_mode == "zeros":
            m.__class__ = BiasSumConv2d
            n += 1
    return n


class _ZeroBiasGrad(Function):
    """identity on z that ties `bias` into the graph with an all-zero gradient"""

    @staticmethod
    def forward(ctx, z, bias):
        ctx.n = bias.shape[0]
        return z.view_as(z)

    @staticmethod
    def backward(ctx, g):
        return g, (g.new_zeros((ctx.n,)) if ctx.needs_input_grad[1] else None)


def cancelled_bias_ready(conv, bn, x):
    """may `bn(conv(x))` drop the convolution's bias?  A bias, a TRAINING-mode BatchNorm2d with running statistics and a momentum,
    a channels-last fp32 device input."""
    return (_ON and isinstance(conv, nn.Conv2d) and conv.bias is not None and conv.padding_mode == "zeros" and isinstance(bn, nn.BatchNorm2d)
            and bn.training and bn.track_running_stats and bn.running_mean is not None and bn.momentum is not None
            and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and _nhwc(x) and x.numel() > 0)


def conv_bn_cancelled_bias(conv, bn, x, bn_call):
    """bn_call(z) for z = conv(x) WITHOUT its bias (see the module docstring); bn_call(z[, statistics rows]) runs the norm (+ activation) on z and updates
    bn.running_mean with mean(z), which is then moved by momentum * bias.  Only in front of the fused norm kernels (bn_act.eligible):
    they update the running statistics through raw pointers and save neither of them, the library's BatchNorm saves both for its
    backward (a later in-place update raises) -- there the convolution keeps its bias."""
    from . import bn_act as BA
    from . import gemm_bf6 as G
    part = None
    if G.eligible(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups):
        if G._STATS and BA._channels_ok(conv.out_channels) and BA._state["enabled"] and type(bn) in BA._BN_TYPES and bn.affine:
            z, part = G.conv1x1_stats(x, conv.weight)               # the norm's statistics from the GEMM's epilogue
        else:
            z = G.conv1x1(x, conv.weight) if torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad) \
                else G.product(x, conv.weight)
    else:
        z = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    if not BA.eligible(z, bn):
        return bn_call(z.add_(conv.bias.view(1, -1, 1, 1)))          # conv(x) as ATen computes it
    if torch.is_grad_enabled() and conv.bias.requires_grad:
        z = _ZeroBiasGrad.apply(z, conv.bias)
    y = bn_call(z) if part is None else bn_call(z, part)
    with torch.no_grad():
        bn.running_mean.add_(conv.bias.detach(), alpha=float(bn.momentum))
    return y
