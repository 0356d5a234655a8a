"""The ResNet stem convolution nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False) (mmdet ResNet._make_stem_layer,
`conv1`; mmdet3d/models/detectors/bevdet.py image_encoder reaches it with the 6-camera batch) on the fp32 matrix cores
(csrc/stem.hip): forward with the statistics of the norm behind it in the epilogue, weight gradient as a persistent fixed-order
kernel.  `StemConv2d` keeps the module's parameters and state-dict keys; anything the kernels do not take (other geometry, NCHW
memory, a bias, an input that wants a gradient) runs the module's own convolution."""
import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L

_ON = os.environ.get("DBEV_STEM_CONV", "1") != "0"


def _nhwc(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def eligible(conv, x):
    return (_ON and isinstance(conv, nn.Conv2d) and conv.in_channels == 3 and conv.out_channels == 64 and conv.kernel_size == (7, 7)
            and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.bias is None and conv.padding_mode == "zeros" and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.shape[1] == 3 and _nhwc(x) and conv.weight.dtype == torch.float32 and x.numel() > 0
            and not (x.requires_grad and torch.is_grad_enabled())
            and L.lib().dbev_stem7x7s2_workspace_bytes(int(x.shape[0]), int(x.shape[2]), int(x.shape[3])) > 0)


class _StemConv(Function):
    @staticmethod
    def forward(ctx, x, weight, stats):
        dev = L.require_cuda(x, weight)
        N, _, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        w = weight if _nhwc(weight) else weight.contiguous(memory_format=torch.channels_last)    # memory [64][7][7][3]
        z = torch.empty((N, 64, Ho, Wo), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        ws = torch.empty((int(L.lib().dbev_stem7x7s2_workspace_bytes(N, H, W)),), dtype=torch.uint8, device=dev)
        part = None
        if stats:
            part = torch.empty((int(L.lib().dbev_stem7x7s2_stats_rows(N, H, W)), 2, 64), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.call("dbev_stem7x7s2_forward", L.ptr(x), L.ptr(w), N, H, W, L.ptr(z), L.ptr(part), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
        ctx.save_for_backward(x, weight)
        if stats:
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)
            return z, part
        return z

    @staticmethod
    def backward(ctx, gz, _gpart=None):
        x, weight = ctx.saved_tensors
        if gz is None or not ctx.needs_input_grad[1]:
            return None, None, None
        dev = gz.device
        N, _, H, W = x.shape
        gz = gz.contiguous(memory_format=torch.channels_last)
        gw = torch.empty((64, 3, 7, 7), dtype=torch.float32, device=dev).contiguous(memory_format=torch.channels_last)
        ws = torch.empty((int(L.lib().dbev_stem7x7s2_workspace_bytes(N, H, W)),), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            L.call("dbev_stem7x7s2_backward_weight", L.ptr(x), L.ptr(gz), N, H, W, L.ptr(gw), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
        if not _nhwc(weight):
            gw = gw.contiguous()
        return None, gw, None


def stem_conv(x, weight):
    return _StemConv.apply(x, weight, False)


def stem_conv_stats(x, weight):
    """-> (z, per-workgroup channel sums of z and z^2 for the norm behind the convolution)"""
    return _StemConv.apply(x, weight, True)


class StemConv2d(nn.Conv2d):
    """nn.Conv2d whose forward takes csrc/stem.hip when `eligible` (re-classed in place by `use_stem_convs`)"""

    def forward(self, x):
        if eligible(self, x):
            return stem_conv(x, self.weight)
        return super().forward(x)


def use_stem_convs(model):
    """Re-class every 3 -> 64, 7x7 / stride-2 / padding-3 bias-free nn.Conv2d of `model`; returns how many.  Idempotent."""
    if not _ON:
        return 0
    n = 0
    for m in model.modules():
        if type(m) in (nn.Conv2d, StemConv2d) and m.in_channels == 3 and m.out_channels == 64 and m.kernel_size == (7, 7) \
                and m.stride == (2, 2) and m.padding == (3, 3) and m.dilation == (1, 1) and m.groups == 1 and m.bias is None \
                and m.padding_mode == "zeros":
            m.__class__ = StemConv2d
            n += 1
    return n
