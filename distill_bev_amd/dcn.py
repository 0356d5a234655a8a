"""DCNv2 (modulated deformable convolution) on the gfx950 sampling kernels.

Mirror of mmcv-full 1.6.0 ``ModulatedDeformConv2dPack`` / ``modulated_deform_conv2d`` as the reference
uses it (registered conv type ``'DCNv2'``; mmdet3d/models/necks/view_transformer_mine.py:298-306,325-329).
The bilinear sampling + modulation (im2col) and its backward (col2im, offset/mask gradients) run in
``libdbev_hip.so`` (``csrc/dcn.hip``); the (k, c) contraction with the weight is a 1x1 convolution of the
channels-last column tensor, i.e. an MFMA GEMM in MIOpen -- the same split as mmcv's im2col + GEMM.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib as L
from .registry import register_conv


class _DCNv2Columns(Function):
    """cols[N, K*C, Ho, Wo] (channels-last) = sigmoid(logit_k) * bilinear(x, p_k + offset_k)."""

    @staticmethod
    def forward(ctx, x, offset_mask, kh, kw, stride, padding, dilation):
        dev = L.require_cuda(x, offset_mask)
        N, C, H, W = x.shape
        K = kh * kw
        Ho = (H + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
        Wo = (W + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
        if tuple(offset_mask.shape) != (N, 3 * K, Ho, Wo):
            raise ValueError(f"offset/mask tensor {tuple(offset_mask.shape)} != {(N, 3 * K, Ho, Wo)}")
        x = x.float().contiguous(memory_format=torch.channels_last)
        om = offset_mask.float().contiguous(memory_format=torch.channels_last)
        cols = torch.empty((N, K * C, Ho, Wo), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        with torch.cuda.device(dev):
            L.call("dbev_dcnv2_im2col", L.ptr(x), L.ptr(om), L.ptr(cols), N, C, H, W, Ho, Wo, kh, kw, stride,
                   padding, dilation, L.stream_ptr(dev))
        ctx.save_for_backward(x, om)
        ctx.dims = (N, C, H, W, Ho, Wo, kh, kw, stride, padding, dilation)
        return cols

    @staticmethod
    def backward(ctx, gcols):
        x, om = ctx.saved_tensors
        N, C, H, W, Ho, Wo, kh, kw, stride, padding, dilation = ctx.dims
        dev = gcols.device
        gcols = gcols.contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(x)          # channels-last; zero-filled by the library
        gom = torch.empty_like(om)
        with torch.cuda.device(dev):
            nbytes = L.call("dbev_dcnv2_col2im_workspace_bytes", N, C, H, W, Ho, Wo, kh, kw)
            ws = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=dev)
            L.call("dbev_dcnv2_col2im", L.ptr(gcols), L.ptr(x), L.ptr(om), L.ptr(gx), L.ptr(gom), N, C, H, W, Ho, Wo,
                   kh, kw, stride, padding, dilation, L.ptr(ws), ws.numel(), L.stream_ptr(dev))
        return gx, gom, None, None, None, None, None


def modulated_deform_conv2d_raw(x, offset_mask, weight, bias, stride=1, padding=1, dilation=1):
    """DCNv2 from the RAW conv_offset output (channels [0,2K) = (dy,dx) pairs, [2K,3K) = mask logits)."""
    Co, C, kh, kw = weight.shape
    cols = _DCNv2Columns.apply(x, offset_mask, kh, kw, stride, padding, dilation)
    w = weight.permute(0, 2, 3, 1).reshape(Co, kh * kw * C, 1, 1)       # (k, c) order of the columns
    if os.environ.get("DBEV_DCN_GEMM") == "1":                           # the same contraction as a plain GEMM (debugging / A-B aid)
        N, KC, Ho, Wo = cols.shape
        y = F.linear(cols.permute(0, 2, 3, 1).reshape(-1, KC), w.view(Co, KC), bias)
        return y.view(N, Ho, Wo, Co).permute(0, 3, 1, 2)
    from . import gemm_bf6 as G
    if G.eligible(cols, w):                                               # the contraction over (k, c): a 1x1 layer of the bf16x6 GEMM
        if torch.is_grad_enabled() and (cols.requires_grad or w.requires_grad or (bias is not None and bias.requires_grad)):
            return G.conv1x1(cols, w, bias)
        y = G.product(cols, w)
        return y if bias is None else y.add_(bias.view(1, -1, 1, 1))
    return F.conv2d(cols, w, bias)


def modulated_deform_conv2d(x, offset, mask, weight, bias, stride=1, padding=1, dilation=1):
    """mmcv call surface: offset[N,2K,Ho,Wo], mask[N,K,Ho,Wo] already squashed to (0,1)."""
    m = mask.clamp(1e-7, 1 - 1e-7)
    return modulated_deform_conv2d_raw(x, torch.cat((offset, torch.log(m) - torch.log1p(-m)), 1), weight, bias,
                                       stride, padding, dilation)


class ModulatedDeformConv2dPack(nn.Module):
    """mmcv ModulatedDeformConv2dPack ('DCNv2'): conv_offset predicts 2K offsets + K mask logits
    (state-dict keys weight, bias, conv_offset.weight, conv_offset.bias as in mmcv)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deform_groups=1, bias=True):
        super().__init__()
        assert groups == 1 and deform_groups == 1
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        self.stride, self.padding, self.dilation, self.k = stride, padding, dilation, k
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, k, k))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.conv_offset = nn.Conv2d(in_channels, 3 * k * k, kernel_size=k, stride=stride, padding=padding,
                                     dilation=dilation, bias=True)
        n = in_channels * k * k
        stdv = 1.0 / n ** 0.5
        nn.init.uniform_(self.weight, -stdv, stdv)
        nn.init.zeros_(self.conv_offset.weight); nn.init.zeros_(self.conv_offset.bias)

    def forward(self, x):
        # mmcv: o1, o2, mask = chunk(out, 3); offset = cat(o1, o2); mask = sigmoid(mask) -- cat(o1, o2) is the
        # first 2K channels of `out` unchanged, and the sigmoid is fused into the sampling kernel
        return modulated_deform_conv2d_raw(x, self.conv_offset(x), self.weight, self.bias, self.stride,
                                           self.padding, self.dilation)


register_conv("DCNv2", ModulatedDeformConv2dPack)
