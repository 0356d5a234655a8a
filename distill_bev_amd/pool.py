"""MaxPool2d(3, stride 2, padding 1) of the ResNet stem on channels-last tensors (csrc/maxpool.hip): ATen's values and tie rule, the
backward as a gather (no atomics).  `max_pool(module, x)` takes the kernels when the module and the tensor fit them, the module
otherwise."""
import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L


_BWD_FUSE = os.environ.get("DBEV_STEM_BWD_FUSE", "1") != "0"


class _MaxPool3x3s2(Function):
    @staticmethod
    def forward(ctx, x):
        dev = x.device
        N, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        win = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            L.call("dbev_maxpool3x3s2_forward", L.ptr(x), N, H, W, C, L.ptr(y), L.ptr(win), L.stream_ptr(dev))
        ctx.save_for_backward(win)
        ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        (win,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = torch.empty((N, C, H, W), dtype=torch.float32, device=gy.device, memory_format=torch.channels_last)
        with torch.cuda.device(gy.device):
            L.call("dbev_maxpool3x3s2_backward", L.ptr(gy), L.ptr(win), N, H, W, C, L.ptr(gx), L.stream_ptr(gy.device))
        return gx


class _NormReluMaxPool(Function):
    """maxpool3x3s2(relu(batch_norm(x))) of the ResNet stem in training mode (mmdet ResNet.forward: conv1 -> norm1 -> relu -> maxpool):
    the statistics pass + finalize of the fused norm (dbev_bn_act_train_forward_mask with y = NULL), then ONE pass that normalises,
    rectifies and pools (dbev_norm_relu_maxpool3x3s2_forward) -- the 554 MB normalised map of the three-module sequence is neither
    written nor read.  Backward: the pooling gather (dbev_maxpool3x3s2_backward), then the fused norm's backward with the ReLU gate
    recomputed from x (dbev_bn_act_backward3, no residual)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, nbt, momentum, eps, part=None):
        # part: per-workgroup channel sums the convolution in front left (stem.stem_conv_stats) -- the statistics pass does not run
        dev = x.device
        N, C, H, W = x.shape
        M = N * H * W
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        save_mean = torch.empty((C,), dtype=torch.float32, device=dev)
        save_invstd = torch.empty((C,), dtype=torch.float32, device=dev)
        coef = torch.empty((2 * C,), dtype=torch.float32, device=dev)
        ws = torch.empty((L.lib().dbev_bn_act_workspace_bytes(M, C) + 12 * C,), dtype=torch.uint8, device=dev)
        y = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        win = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            L.call("dbev_bn_act_train_forward_mask", L.ptr(x), None, L.ptr(weight), L.ptr(bias), L.ptr(running_mean), L.ptr(running_var),
                   L.ptr(nbt), float(momentum or 0.0), float(eps), 1, None, L.ptr(save_mean), L.ptr(save_invstd), L.ptr(coef), M, C,
                   L.ptr(part), 0 if part is None else int(part.shape[0]), None, L.ptr(ws), ws.numel(), L.stream_ptr(dev))
            L.call("dbev_norm_relu_maxpool3x3s2_forward", L.ptr(x), L.ptr(coef), N, H, W, C, L.ptr(y), L.ptr(win), L.stream_ptr(dev))
        L.touched(running_mean, running_var, nbt)
        ctx.save_for_backward(x, weight, save_mean, save_invstd, coef, win)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, save_mean, save_invstd, coef, win = ctx.saved_tensors
        N, C, H, W = x.shape
        M = N * H * W
        dev = gy.device
        gy = gy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
        nb = int(L.lib().dbev_stem_pool_norm_backward_workspace_bytes(N, H, W, C)) if _BWD_FUSE else 0
        if nb > 0:                               # the pooling gather inside both passes of the norm's backward (no 554 MB gradient map)
            ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                L.call("dbev_stem_pool_norm_backward", L.ptr(gy), L.ptr(win), L.ptr(x), L.ptr(weight), L.ptr(save_mean), L.ptr(save_invstd),
                       L.ptr(coef), N, H, W, C, L.ptr(dx), L.ptr(dgamma), L.ptr(dbeta), L.ptr(ws), nb, L.stream_ptr(dev))
            return dx, dgamma, dbeta, None, None, None, None, None, None
        ga = torch.empty_like(x)
        ws = torch.empty((L.lib().dbev_bn_act_workspace_bytes(M, C) + 12 * C,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            L.call("dbev_maxpool3x3s2_backward", L.ptr(gy), L.ptr(win), N, H, W, C, L.ptr(ga), L.stream_ptr(dev))
            L.call("dbev_bn_act_backward3", L.ptr(ga), None, L.ptr(x), None, 0, L.ptr(weight), L.ptr(save_mean), L.ptr(save_invstd),
                   L.ptr(coef), 1, L.ptr(dx), None, L.ptr(dgamma), L.ptr(dbeta), M, C, L.ptr(ws), ws.numel(), L.stream_ptr(dev))
        return dx, dgamma, dbeta, None, None, None, None, None, None


def _pool_geometry(module):
    return (type(module) is nn.MaxPool2d and _pair(module.kernel_size) == (3, 3) and _pair(module.stride) == (2, 2)
            and _pair(module.padding) == (1, 1) and _pair(module.dilation) == (1, 1) and not module.ceil_mode
            and not module.return_indices)


def _fusable(norm, pool, x):
    from . import bn_act as BA
    return (os.environ.get("DBEV_STEM_FUSE", "1") != "0" and _pool_geometry(pool) and BA.eligible(x, norm) and norm.training
            and norm.running_mean is not None and norm.momentum is not None)


def norm_relu_max_pool(norm, pool, x, part=None):
    """pool(relu(norm(x))) -- one fused pass when `norm` is a training-mode BatchNorm2d the fused norm kernels take and `pool` the
    stem's 3x3 / stride-2 pooling; the modules' own sequence (through bn_act / max_pool) otherwise"""
    from . import bn_act as BA
    if _fusable(norm, pool, x):
        return _NormReluMaxPool.apply(x, norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.num_batches_tracked,
                                      norm.momentum, norm.eps, part)
    return max_pool(pool, BA.bn_act(x, norm, None, True))


def conv_norm_relu_max_pool(conv, norm, pool, x):
    """The whole stem, pool(relu(norm(conv(x)))) (mmdet ResNet.forward): the 7x7 convolution on csrc/stem.hip with the norm's
    statistics in its epilogue when the module and the tensors fit (stem.StemConv2d + a training-mode norm), `conv(x)` otherwise"""
    from . import stem as S
    if isinstance(conv, S.StemConv2d) and S.eligible(conv, x) and norm.training and os.environ.get("DBEV_STEM_STATS", "1") != "0":
        z, part = S.stem_conv_stats(x, conv.weight)
        return norm_relu_max_pool(norm, pool, z, part if _fusable(norm, pool, z) else None)
    return norm_relu_max_pool(norm, pool, conv(x))


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def max_pool(module, x):
    if (type(module) is nn.MaxPool2d and _pair(module.kernel_size) == (3, 3) and _pair(module.stride) == (2, 2)
            and _pair(module.padding) == (1, 1) and _pair(module.dilation) == (1, 1) and not module.ceil_mode
            and not module.return_indices and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] % 4 == 0
            and x.is_contiguous(memory_format=torch.channels_last) and x.numel() > 0):
        return _MaxPool3x3s2.apply(x)
    return module(x)
