"""MaxPool2d(3, stride 2, padding 1) of the ResNet stem on channels-last tensors (csrc/maxpool.hip): ATen's values and tie rule, the
backward as a gather (no atomics).  `max_pool(module, x)` takes the kernels when the module and the tensor fit them, the module
otherwise."""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L


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


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def max_pool(module, x):
    if (type(module) is nn.MaxPool2d and _pair(module.kernel_size) == (3, 3) and _pair(module.stride) == (2, 2)
            and _pair(module.padding) == (1, 1) and _pair(module.dilation) == (1, 1) and not module.ceil_mode
            and not module.return_indices and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] % 4 == 0
            and x.is_contiguous(memory_format=torch.channels_last) and x.numel() > 0):
        return _MaxPool3x3s2.apply(x)
    return module(x)
