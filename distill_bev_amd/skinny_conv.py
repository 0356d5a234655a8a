"""3x3 / stride 1 / padding 1 convolutions with 1-3 output channels on the gfx950 streaming kernels
(csrc/skinny_conv.hip): the final layer of every CenterHead branch (SeparateHead, centerpoint_head.py:17-130).

`SkinnyConv2d` is an nn.Conv2d (same parameters, same state-dict keys); `use_skinny_convs(model)` re-classes the
matching nn.Conv2d modules of a built model.  Inputs the kernels do not cover (CPU, NCHW-contiguous, other
geometries) take nn.Conv2d's own forward (MIOpen).
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L

_OK_C4 = (8, 16, 32, 64)


def _nhwc(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def eligible(x, weight):
    co, ci, kh, kw = weight.shape
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and _nhwc(x) and kh == 3 and kw == 3
            and 1 <= co <= 3 and ci % 4 == 0 and (ci // 4) in _OK_C4 and x.shape[1] == ci and x.numel() > 0)


class _SkinnyConv3x3(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        dev = x.device
        N, Ci, H, W = x.shape
        Co = weight.shape[0]
        wp = weight.permute(0, 2, 3, 1).contiguous()                  # [Co, 3, 3, Ci]
        y = torch.empty((N, Co, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        with torch.cuda.device(dev):
            L.call("dbev_skinny_conv3x3_forward", L.ptr(x), L.ptr(wp), L.ptr(bias), L.ptr(y), N, Ci, H, W, Co,
                   L.stream_ptr(dev), alg_bytes=4 * N * H * W * (Ci + Co))
        ctx.save_for_backward(x, wp)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, wp = ctx.saved_tensors
        dev = gy.device
        N, Ci, H, W = x.shape
        Co = wp.shape[0]
        # [N, Co, H, W] channels-last == [N, H, W, Co]; for Co == 1 any dense layout already is
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        gwp = torch.empty_like(wp) if need_w else None
        gb = torch.empty((Co,), dtype=torch.float32, device=dev) if need_w else None
        with torch.cuda.device(dev):
            nbytes = L.call("dbev_skinny_conv3x3_workspace_bytes", Ci, Co)
            ws = torch.empty((int(nbytes),), dtype=torch.uint8, device=dev) if need_w else None
            L.call("dbev_skinny_conv3x3_backward", L.ptr(gy), L.ptr(x), L.ptr(wp), L.ptr(gx), L.ptr(gwp), L.ptr(gb), N, Ci,
                   H, W, Co, L.ptr(ws), ws.numel() if ws is not None else 0, L.stream_ptr(dev),
                   alg_bytes=4 * N * H * W * (2 * Ci + 2 * Co))
        gw = gwp.permute(0, 3, 1, 2) if gwp is not None else None     # back to [Co, Ci, 3, 3] (a view)
        return gx, gw, (gb if ctx.has_bias else None)


def skinny_conv3x3(x, weight, bias=None):
    """F.conv2d(x, weight, bias, stride=1, padding=1) for 3x3 kernels with <= 3 output channels, NHWC fp32 on the GPU."""
    return _SkinnyConv3x3.apply(x, weight, bias)


class SkinnyConv2d(nn.Conv2d):
    def forward(self, x):
        if (self.stride == (1, 1) and self.padding == (1, 1) and self.dilation == (1, 1) and self.groups == 1
                and self.padding_mode == "zeros" and eligible(x, self.weight)):
            return skinny_conv3x3(x, self.weight, self.bias)
        if x.is_cuda and x.numel() > 0:
            L.note_fallback("skinny_conv", "not channels-last" if not _nhwc(x) else f"geometry {tuple(self.weight.shape)}")
        return super().forward(x)


def use_skinny_convs(model):
    """Re-class the nn.Conv2d modules with the skinny geometry (3x3, s1, p1, <= 3 output channels, Cin/4 in
    {8,16,32,64}); returns how many.  Parameters and state-dict keys are untouched.  Idempotent."""
    n = 0
    for m in model.modules():
        if type(m) is nn.Conv2d and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) \
                and m.dilation == (1, 1) and m.groups == 1 and m.padding_mode == "zeros" and 1 <= m.out_channels <= 3 \
                and m.in_channels % 4 == 0 and (m.in_channels // 4) in _OK_C4:
            m.__class__ = SkinnyConv2d
            n += 1
    return n
