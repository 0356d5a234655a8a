"""1x1 convolutions on the fp32-MFMA GEMM kernels without VALU work in their main loops (csrc/gemm1x1.hip).

Host-side mirror of what the reference reaches through ``nn.Conv2d(k=1)`` -> cuDNN in the bottlenecks and necks
(mmdet3d/models/bricks/res_block.py:102-230, necks/fpn.py:10-204, necks/lss_fpn.py:10-72): ``conv1x1(x, weight)`` is
``F.conv2d(x, weight)`` for channels-last fp32 device tensors with N*H*W % 128 == 0, Cin % 32 == 0, Cout % 64 == 0, differentiable:
the data gradient is the same kernel on grad_y with the transposed weight, the weight gradient ``dbev_gemm1x1_backward_weight``
(the library's kernels where a channel count does not fit).  ``conv1x1_stats`` also returns the partial (sum y, sum y^2) rows of the
output for the fused BatchNorm that follows (bn_act(..., pre=rows))."""
import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L

# Opt-in (DBEV_GEMM1X1=1): measured at parity with MIOpen's tuned implicit GEMM on the step's 1x1 layers (forward 0.67-1.12 x, data
# gradient 0.65-1.44 x, weight gradient 0.93-1.30 x: profiles/r04_gemm1x1_vs_miopen.txt) -- both sit at 75-85 % of the fp32 MFMA peak, so
# the bench configuration keeps the library's kernels for them; the module surface and the kernels stay tested.
_ON = os.environ.get("DBEV_GEMM1X1", "0") == "1"


def _nhwc(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def eligible(x, weight, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1):
    """can `F.conv2d(x, weight)` run on the GEMM kernels? (geometry only; GemmConv2d / use_gemm_convs also look at the opt-in switch)"""
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and weight.dim() == 4):
        return False
    if tuple(weight.shape[2:]) != (1, 1) or tuple(stride) != (1, 1) or tuple(padding) != (0, 0) or tuple(dilation) != (1, 1) or groups != 1:
        return False
    N, C, H, W = x.shape
    Co = weight.shape[0]
    return weight.shape[1] == C and (N * H * W) % 128 == 0 and C % 32 == 0 and Co % 64 == 0 and _nhwc(x)


def _gemm(x2d_ptr_owner, M, Cin, Cout, w2d, stats, dev, out_shape):
    """y[M, Cout] = x[M, Cin] @ w2d[Cout, Cin]^T through dbev_gemm1x1_forward; x2d_ptr_owner: a channels-last [N, Cin, H, W] tensor"""
    y = torch.empty(out_shape, dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    part = None
    if stats:
        rows = int(L.call("dbev_gemm1x1_stats_rows", M, Cin, Cout, Cin))
        part = torch.empty((rows, 2, Cout), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        L.call("dbev_gemm1x1_forward", L.ptr(x2d_ptr_owner), L.ptr(w2d), L.ptr(y), L.ptr(part), M, Cin, Cout, Cin, L.stream_ptr(dev))
    return y, part


class _Conv1x1(Function):
    @staticmethod
    def forward(ctx, x, weight, stats):
        dev = L.require_cuda(x, weight)
        N, C, H, W = x.shape
        Co = weight.shape[0]
        w2 = weight.reshape(Co, C)
        w2 = w2 if w2.is_contiguous() else w2.contiguous()
        y, part = _gemm(x, N * H * W, C, Co, w2, stats, dev, (N, Co, H, W))
        ctx.save_for_backward(x, weight)
        if stats:
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)              # no zero tensor for the (never used) gradient of the statistics output
            return y, part
        return y

    @staticmethod
    def backward(ctx, gy, _gpart=None):
        x, weight = ctx.saved_tensors
        if gy is None:
            return None, None, None
        gy = gy.contiguous(memory_format=torch.channels_last)
        dev = gy.device
        N, C, H, W = x.shape
        Co = weight.shape[0]
        M = N * H * W
        gx = gw = None
        lib_mask = [False, False, False]
        if ctx.needs_input_grad[0]:
            if Co % 32 == 0 and C % 64 == 0:               # the forward kernel on grad_y with the transposed weight [C, Co]
                wt = weight.reshape(Co, C).t().contiguous()
                gx, _ = _gemm(gy, M, Co, C, wt, False, dev, (N, C, H, W))
            else:
                lib_mask[0] = True
        if ctx.needs_input_grad[1]:
            nbytes = int(L.call("dbev_gemm1x1_backward_weight_workspace_bytes", M, C, Co, C)) if (C % 64 == 0 and Co % 64 == 0) else 0
            if nbytes:
                g2 = torch.empty((Co, C), dtype=torch.float32, device=dev)
                ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
                with torch.cuda.device(dev):
                    L.call("dbev_gemm1x1_backward_weight", L.ptr(x), L.ptr(gy), L.ptr(g2), M, C, Co, C, L.ptr(ws), nbytes, L.stream_ptr(dev))
                gw = torch.empty_like(weight)              # the parameter's own strides (a 1x1 kernel: any layout is [Co, C] in memory)
                gw.view(-1)[:] = g2.view(-1) if gw.is_contiguous() or gw.is_contiguous(memory_format=torch.channels_last) else g2.reshape(weight.shape).reshape(-1)
            else:
                lib_mask[1] = True
        if any(lib_mask):
            a, b, _ = torch.ops.aten.convolution_backward(gy, x, weight, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, lib_mask)
            gx = a if lib_mask[0] else gx
            gw = b if lib_mask[1] else gw
        return gx, gw, None


def conv1x1(x, weight):
    """F.conv2d(x, weight) for an `eligible` pair"""
    return _Conv1x1.apply(x, weight, False)


def conv1x1_stats(x, weight):
    return _Conv1x1.apply(x, weight, True)


class GemmConv2d(nn.Conv2d):
    """nn.Conv2d (1x1, stride 1, no padding, no bias) whose forward, data gradient and weight gradient run on the GEMM kernels when the
    input qualifies (`eligible`); the stock convolution otherwise.  Same parameters and state-dict keys."""

    def forward(self, x):
        if _ON and self.bias is None and eligible(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
            return conv1x1(x, self.weight)
        return super().forward(x)


def use_gemm_convs(model):
    """Re-class the bias-free 1x1 / stride-1 nn.Conv2d modules with Cin % 32 == 0 and Cout % 64 == 0; returns how many.  Idempotent."""
    if not _ON:
        return 0
    n = 0
    for m in model.modules():
        if type(m) is nn.Conv2d and m.kernel_size == (1, 1) and m.stride == (1, 1) and m.padding == (0, 0) and m.dilation == (1, 1) \
                and m.groups == 1 and m.bias is None and m.in_channels % 32 == 0 and m.out_channels % 64 == 0:
            m.__class__ = GemmConv2d
            n += 1
    return n
