"""Fused depth head of the BEVDepth view transformer: BatchNorm2d -> 1x1 convolution -> softmax in one pass.

Reference sequence (mmdet3d/models/necks/view_transformer_mine.py:300-309, 325-328; detectors/bevdet_distill_more.py:398-416):

    depth_feat  = self.dcn(depth_feat)              # nn.Sequential(DCNv2, nn.BatchNorm2d(c))
    depth_digit = self.depthnet(depth_feat)         # nn.Conv2d(c, D, kernel_size=1)
    depth_prob  = self.get_depth_dist(depth_digit)  # softmax(dim=1)

`depth_head(x, bn, conv)` takes the deformable convolution's output x and returns (depth_digit, depth_prob), both channels-last
[BN, D, H, W].  Forward: the norm's batch statistics (training) come from the statistics half of the fused norm kernels
(`dbev_bn_act_train_forward_pre` with y = NULL; running statistics and `num_batches_tracked` updated there), then ONE kernel
(`dbev_depth_head_forward`, csrc/depth_head.hip) normalises, multiplies on the fp32 matrix cores, adds the bias and takes the
softmax.  Backward: softmax / bias / 1x1 gradients through ATen on the small [BN, D, H, W] maps and the saved normalised map, the
norm's backward through the fused norm kernel (`dbev_bn_act_backward`).  Anything the kernel does not cover (other dtypes / layouts,
more than 64 depth bins or 256 channels, an eval-mode norm inside autograd) takes the module sequence and is counted in the fallback
ledger.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L
from . import bn_act as BA

_state = {"enabled": True}


def set_enabled(flag):
    _state["enabled"] = bool(flag)


def _conv_ok(conv):
    from .colsum import BiasSumConv2d                       # (the re-classed module only changes the autograd path of conv.forward, which this op replaces)
    return (type(conv) in (nn.Conv2d, BiasSumConv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is not None and conv.out_channels <= 64
            and conv.in_channels <= 256 and conv.weight.dtype == torch.float32)


def eligible(x, bn, conv):
    return bool(_state["enabled"] and _conv_ok(conv) and conv.in_channels == x.shape[1] and BA.eligible(x, bn)
                and not (not bn.training and torch.is_grad_enabled() and (conv.weight.requires_grad or x.requires_grad)))


class _DepthHead(Function):
    @staticmethod
    def forward(ctx, x, bn_w, bn_b, conv_w, conv_b, running_mean, running_var, nbt, momentum, eps, training):
        dev = x.device
        BN, C, H, W = x.shape
        M, N = BN * H * W, conv_w.shape[0]
        coef = torch.empty((2 * C,), dtype=torch.float32, device=dev)
        save_mean = save_invstd = None
        with torch.cuda.device(dev):
            if training:
                save_mean = torch.empty((C,), dtype=torch.float32, device=dev)
                save_invstd = torch.empty((C,), dtype=torch.float32, device=dev)
                ws = torch.empty((L.lib().dbev_bn_act_workspace_bytes(M, C) + 12 * C,), dtype=torch.uint8, device=dev)
                L.call("dbev_bn_act_train_forward_pre", L.ptr(x), L.ptr(None), L.ptr(bn_w), L.ptr(bn_b), L.ptr(running_mean),
                       L.ptr(running_var), L.ptr(nbt), float(momentum or 0.0), float(eps), 0, L.ptr(None), L.ptr(save_mean),
                       L.ptr(save_invstd), L.ptr(coef), M, C, L.ptr(None), 0, L.ptr(ws), ws.numel(), L.stream_ptr(dev),
                       alg_bytes=4 * M * C)
                L.touched(running_mean, running_var, nbt)
            else:
                scale = bn_w * torch.rsqrt(running_var + eps)
                coef[:C] = scale
                coef[C:] = bn_b - running_mean * scale
            need_grad = training and any(ctx.needs_input_grad[:5])      # all False under no_grad (the adjacent frame of BEVDepth4D)
            digit = torch.empty((BN, N, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
            prob = torch.empty_like(digit)
            xn = torch.empty_like(x) if need_grad else None
            w2 = conv_w.reshape(N, C).contiguous()
            L.call("dbev_depth_head_forward", L.ptr(x), L.ptr(coef), L.ptr(w2), L.ptr(conv_b), M, C, N, L.ptr(digit), L.ptr(prob),
                   L.ptr(xn), L.stream_ptr(dev), alg_bytes=4 * M * (C * (2 if need_grad else 1) + 2 * N))
        if need_grad:
            ctx.save_for_backward(x, xn, prob, bn_w, conv_w, save_mean, save_invstd, coef)
        return digit, prob

    @staticmethod
    def backward(ctx, g_digit, g_prob):
        x, xn, prob, bn_w, conv_w, save_mean, save_invstd, coef = ctx.saved_tensors
        dev = x.device
        BN, C, H, W = x.shape
        M = BN * H * W
        # softmax: g_z = g_digit + p * (g_p - sum_j g_p[j] p[j])
        gz = g_digit
        if g_prob is not None:
            t = prob * (g_prob - (g_prob * prob).sum(dim=1, keepdim=True))
            gz = t if gz is None else gz + t
        gz = gz.contiguous(memory_format=torch.channels_last)
        gxn, gw, _ = torch.ops.aten.convolution_backward(gz, xn, conv_w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                         [True, True, False])
        from .colsum import channel_sum
        gb = channel_sum(gz) if ctx.needs_input_grad[4] else None      # ATen's reduction takes ~95 us for this 59-channel map
        gxn = gxn.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
        ws = torch.empty((L.lib().dbev_bn_act_workspace_bytes(M, C) + 12 * C,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            L.call("dbev_bn_act_backward", L.ptr(gxn), L.ptr(x), L.ptr(None), L.ptr(bn_w), L.ptr(save_mean), L.ptr(save_invstd),
                   L.ptr(coef), 0, L.ptr(dx), L.ptr(None), L.ptr(dgamma), L.ptr(dbeta), M, C, L.ptr(ws), ws.numel(),
                   L.stream_ptr(dev), alg_bytes=4 * M * C * 5)
        return dx, dgamma, dbeta, gw, gb, None, None, None, None, None, None


def depth_head(x, bn, conv):
    """-> (depth_digit, depth_prob) = (conv(bn(x)), softmax(conv(bn(x)), dim=1))"""
    if not eligible(x, bn, conv):
        if x.is_cuda:
            L.note_fallback("depth_head", "module sequence (norm -> 1x1 -> softmax)")
        digit = conv(bn(x))
        return digit, digit.softmax(dim=1)
    training = bn.training or bn.running_mean is None
    return _DepthHead.apply(x, bn.weight, bn.bias, conv.weight, conv.bias, bn.running_mean, bn.running_var,
                            bn.num_batches_tracked, bn.momentum, bn.eps, training)
