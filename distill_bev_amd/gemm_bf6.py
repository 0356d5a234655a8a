"""1x1 convolutions as fp32 GEMMs on the BF16 matrix cores at fp32 accuracy (csrc/gemm_bf6.hip, "bf16x6").

Host-side mirror of what the reference reaches through ``nn.Conv2d(k=1)`` -> cuDNN in the bottlenecks and necks
(mmdet3d/models/bricks/res_block.py:102-230, necks/fpn.py:10-204, necks/lss_fpn.py:10-72).  Every fp32 operand is split into three
bf16 values (24 mantissa bits in three pieces), a product is the sum of the six partial products that matter -- each exact in fp32 --
and the sums are kept in fp32: max |y - fp64| / max |y| = 2-4e-7 on the step's layers, at or BELOW the library's fp32 kernels
(5e-7-1e-6; tests/test_gpu_gemm_bf6.py asserts it per shape).  On gfx950 the bf16 matrix pipe is 16 x as fast as the fp32 one, so six
bf16 instructions replace eight fp32 ones in 0.38 of the time.

``conv1x1(x, weight)`` is ``F.conv2d(x, weight)`` for channels-last fp32 device tensors with Cin % 64 == 0,
Cout % 64 == 0, differentiable: the data gradient is the same kernel on grad_y with the transposed weight view, the weight gradient
its own kernel (both operands split on the fly).  Layers that would leave half of the chip idle (fewer than `_MIN_ITEMS` output tiles) stay with the library."""
import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L

L.ensure_param_version_hook()          # fused optimizers do not move `_version`; the kept packs / coefficients follow it

_ON = os.environ.get("DBEV_BF6", "1") != "0"
_WGRAD = os.environ.get("DBEV_BF6_WGRAD", "1") != "0"
_STATS = os.environ.get("DBEV_BF6_STATS", "1") != "0"         # BatchNorm statistics from the forward kernel's epilogue (nets._conv1x1_stats)
_MIN_ITEMS = int(os.environ.get("DBEV_BF6_MIN_ITEMS", "448"))
_WGRAD_LIB_ROWS = int(os.environ.get("DBEV_BF6_WGRAD_LIB_ROWS", "262144"))
_MIN_WGRAD_ROWS = 4096                                           # below: a handful of chunks per share, the library's kernel     # 128 x 128 (or 128 x 64) output tiles; two workgroups share a CU


def _nhwc(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def tile_n(M, N):
    """columns of a workgroup tile for Y[M, N]: 128 when that already gives `_MIN_ITEMS` tiles (or N % 128 != 0 -> 64), else 64 -- the
    8 x 22 maps of the last ResNet stage have 66 row blocks"""
    if N % 128:
        return 64
    return 128 if -(-M // 128) * (N // 128) >= _MIN_ITEMS else 64


def shape_ok(M, K, N):
    """can the kernels take Y[M, N] = X[M, K] W[N, K]^T, and does the launch fill the chip?  (M: any row count since round 6 -- the
    last 128-row block is read and written through bounds-checked buffer descriptors)"""
    if M <= 0 or K % 64 or N % 64 or K <= 0 or N <= 0:
        return False
    return -(-M // 128) * (N // tile_n(M, N)) >= _MIN_ITEMS


def eligible(x, weight, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1):
    if not (_ON and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and weight.dim() == 4):
        return False
    if tuple(weight.shape[2:]) != (1, 1) or tuple(stride) != (1, 1) or tuple(padding) != (0, 0) or tuple(dilation) != (1, 1) or groups != 1:
        return False
    N, C, H, W = x.shape
    return weight.shape[1] == C and _nhwc(x) and shape_ok(N * H * W, C, weight.shape[0])


def matrix(weight):
    """the [Cout, K] matrix the GEMM kernels multiply by: a 1x1 filter itself (either memory format), or the channels-last memory
    [Cout][ky][kx][c] of a k x k filter taken as K = k k Cin columns (a view when the filter is channels-last contiguous)"""
    if weight.dim() == 2:                                    # nn.Linear's [out_features, in_features]
        return weight.detach()
    Co, Ci, kh, kw = weight.shape
    if (kh, kw) == (1, 1):
        return weight.detach().reshape(Co, Ci)
    return weight.detach().permute(0, 2, 3, 1).reshape(Co, kh * kw * Ci)


def packed(weight, transposed=False, tn=128):
    """the three bf16 planes of `weight` [Cout, Cin, 1, 1] in the kernel's LDS image order (dbev_gemm_bf16x6_pack), kept on the weight
    until it changes.  transposed: for the data gradient (rows = input channels, reduction over the output channels)."""
    dev = L.require_cuda(weight)
    w2 = matrix(weight)                                       # a view for both memory formats of a 1x1 filter
    Co, Ci = int(w2.shape[0]), int(w2.shape[1])
    key = (weight._version, weight.data_ptr())
    cache = _cache_for(weight, key)
    if (Ci if transposed else Co) % 128:
        tn = 64
    which = (bool(transposed), tn)
    hit = cache[1].get(which)
    if hit is not None:
        L.note_derived("bf6", weight, which, (hit,))
        return hit
    n, k = (Ci, Co) if transposed else (Co, Ci)
    sn, sk = (w2.stride(1), w2.stride(0)) if transposed else (w2.stride(0), w2.stride(1))
    nbytes = int(L.call("dbev_gemm_bf16x6_packed_bytes", n, k))
    if nbytes == 0:
        raise L.DbevHipError(f"gemm_bf6: unsupported weight {Co} x {Ci} (transposed={transposed})")
    buf = _spare(cache, which, nbytes, dev)
    with torch.cuda.device(dev):
        L.call("dbev_gemm_bf16x6_pack", L.ptr(w2), sn, sk, n, k, tn, L.ptr(buf), L.stream_ptr(dev))
    cache[1][which] = buf
    L.note_derived("bf6", weight, which, (buf,))
    return buf


def _cache_for(weight, key):
    """the weight's pack cache (key, {(transposed, tile width): planes}, fingerprint, spare buffers) for its CURRENT version.  When the
    version moved, the stale entry's buffers become the new entry's spares: the next pack of the same kind is written into the same
    buffer (`_spare`), so the address a captured hipGraph baked in (graphed.py) keeps pointing at fresh planes."""
    cache = getattr(weight, "_dbev_bf6_packs", None)
    if cache is None or cache[0] != key:
        spare = {}
        if cache is not None and cache[0][1] == key[1]:
            spare = dict(cache[3]) if len(cache) > 3 else {}
            spare.update(cache[1])
        cache = (key, {}, L.fingerprint(weight), spare)
        try:
            weight._dbev_bf6_packs = cache
        except AttributeError:
            pass
    else:
        L.check_fingerprint(cache[2], "bf16 planes of a 1x1 filter", weight)
    return cache


def _spare(cache, which, nbytes, dev):
    buf = cache[3].pop(which, None) if len(cache) > 3 else None
    if buf is None or buf.numel() != nbytes or buf.device != torch.device(dev):
        buf = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    return buf


def pack_both(weight, M):
    """forward and data-gradient planes of `weight` for a layer of M rows in ONE launch (dbev_gemm_bf16x6_pack_pair), into the cache
    `packed` reads -- called where both will be needed (a differentiated forward whose data gradient the kernels take)"""
    Co, Ci = int(weight.shape[0]), int(weight.shape[1])
    tf, tt = tile_n(M, Co), tile_n(M, Ci)
    key = (weight._version, weight.data_ptr())
    cache = getattr(weight, "_dbev_bf6_packs", None)
    if cache is not None and cache[0] == key and (False, tf) in cache[1] and (True, tt) in cache[1]:
        return
    cache = _cache_for(weight, key)
    if getattr(weight, "_dbev_bf6_packs", None) is not cache:
        return                                                # no place to keep them: `packed` packs one at a time
    dev = L.require_cuda(weight)
    w2 = weight.detach().reshape(Co, Ci)
    nf, nt = int(L.call("dbev_gemm_bf16x6_packed_bytes", Co, Ci)), int(L.call("dbev_gemm_bf16x6_packed_bytes", Ci, Co))
    if nf == 0 or nt == 0:
        return
    bf = cache[1].get((False, tf))
    bt = cache[1].get((True, tt))
    bf = bf if bf is not None else _spare(cache, (False, tf), nf, dev)
    bt = bt if bt is not None else _spare(cache, (True, tt), nt, dev)
    with torch.cuda.device(dev):
        L.call("dbev_gemm_bf16x6_pack_pair", L.ptr(w2), w2.stride(0), w2.stride(1), Co, Ci, tf, L.ptr(bf), tt, L.ptr(bt), L.stream_ptr(dev))
    cache[1][(False, tf)] = bf
    cache[1][(True, tt)] = bt


def gemm(x, pack, Cout, tn, stats=False, bias=None):
    """x [N, K, H, W] channels-last -> [N, Cout, H, W] channels-last with weight planes packed for tile width `tn` (`product` pairs
    the pack and the launch); stats: also the partial BatchNorm statistics rows f32[M / 128, 2, Cout] of the output (kernel epilogue)"""
    dev = L.require_cuda(x, pack)
    n, K, H, W = x.shape
    M = n * H * W
    y = torch.empty((n, Cout, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    part = torch.empty((int(L.call("dbev_gemm_bf16x6_stats_rows", M)), 2, Cout), dtype=torch.float32, device=dev) if stats else None
    with torch.cuda.device(dev):
        if bias is not None and not stats:                    # (round 6: the bias in the kernel's epilogue)
            L.call("dbev_gemm_bf16x6_forward_bias", L.ptr(x), L.ptr(pack), L.ptr(bias), L.ptr(y), M, K, Cout, K, int(tn), L.stream_ptr(dev))
        else:
            L.call("dbev_gemm_bf16x6_forward_stats", L.ptr(x), L.ptr(pack), L.ptr(y), L.ptr(part), M, K, Cout, K, int(tn), L.stream_ptr(dev))
    return (y, part) if stats else y


def product(x, weight, transposed=False, stats=False, bias=None):
    """x [N, K, H, W] channels-last times the 1x1 filter `weight` [Cout, Cin, 1, 1] (transposed: its transpose, the data gradient's
    operand) -> channels-last; no autograd.  The tile width follows the layer's row count (`tile_n`), the pack is made for it."""
    cout = int(weight.shape[1 if transposed else 0])
    tn = tile_n(x.shape[0] * x.shape[2] * x.shape[3], cout)
    return gemm(x, packed(weight, transposed, tn), cout, tn, stats, bias)


def data_gradient(gy, weight):
    """grad_x of y = conv1x1(x, weight) from a channels-last grad_y, or None when the library should do it (shape / layout)"""
    Co, Ci = int(weight.shape[0]), int(weight.shape[1])
    if not (_ON and gy.is_cuda and gy.dtype == torch.float32 and _nhwc(gy) and shape_ok(gy.shape[0] * gy.shape[2] * gy.shape[3], Co, Ci)):
        return None
    return product(gy, weight, True)


def weight_gradient(x, gy, weight):
    """grad_weight of y = conv1x1(x, weight) (dbev_gemm_bf16x6_backward_weight: both operands split on the fly, fixed summation order),
    with `weight`'s strides, or None when the library should do it (channel counts not multiples of 64, layout)"""
    Co, Ci = int(weight.shape[0]), int(weight.shape[1])
    if not (_ON and _WGRAD and x.is_cuda and x.dtype == torch.float32 and gy.dtype == torch.float32 and _nhwc(x) and _nhwc(gy)
            and Ci % 64 == 0 and Co % 64 == 0):
        return None
    M = x.shape[0] * x.shape[2] * x.shape[3]
    nbytes = int(L.call("dbev_gemm_bf16x6_backward_weight_workspace_bytes", M, Ci, Co, Ci))
    if nbytes == 0 or M < _MIN_WGRAD_ROWS:
        return None
    if min(Ci, Co) <= 64 and M >= _WGRAD_LIB_ROWS:
        # the 64 <-> 256 layers of the first ResNet stage (540 k pixels): HBM-bound, and the library's kernel streams them faster
        # (190 vs 215 us, profiles/r05_gemm_bf6_vs_miopen.txt)
        return None
    dev = x.device
    gw2 = torch.empty((Co, Ci), dtype=torch.float32, device=dev)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.call("dbev_gemm_bf16x6_backward_weight", L.ptr(x), L.ptr(gy), L.ptr(gw2), M, Ci, Co, Ci, L.ptr(ws), nbytes, L.stream_ptr(dev))
    return gw2.view(weight.shape)                           # [Co, Ci] in memory: contiguous in both memory formats of a 1x1 filter (and a Linear's 2-D weight)


# ---- stride-2 1x1 convolutions: subsample + GEMM (csrc/stride2.hip) -------------------------------------------------------------------
_S2 = os.environ.get("DBEV_BF6_S2", "1") != "0"


def eligible_s2(x, weight, stride=(2, 2), padding=(0, 0), dilation=(1, 1), groups=1):
    """can `F.conv2d(x, weight, stride=2)` (1x1, no padding) run as subsample + bf16x6 GEMM?  channels-last fp32, H and W even, the
    GEMM's own shape rules on the subsampled pixels"""
    if not (_ON and _S2 and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and weight.dim() == 4):
        return False
    if tuple(weight.shape[2:]) != (1, 1) or tuple(stride) != (2, 2) or tuple(padding) != (0, 0) or tuple(dilation) != (1, 1) or groups != 1:
        return False
    N, C, H, W = x.shape
    return weight.shape[1] == C and H % 2 == 0 and W % 2 == 0 and _nhwc(x) and shape_ok(N * (H // 2) * (W // 2), C, weight.shape[0])


def subsample2(x):
    """x[:, :, ::2, ::2] of a channels-last tensor as a contiguous channels-last tensor (one streaming pass)"""
    dev = L.require_cuda(x)
    N, C, H, W = x.shape
    y = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    with torch.cuda.device(dev):
        L.call("dbev_subsample2_nhwc", L.ptr(x), L.ptr(y), N, H, W, C, L.stream_ptr(dev))
    return y


def upsample2_zero(g, H, W):
    """the data gradient of subsample2: g [N, C, H/2, W/2] channels-last -> [N, C, H, W] with g at the even pixels and zeros elsewhere"""
    dev = L.require_cuda(g)
    N, C = g.shape[:2]
    gx = torch.empty((N, C, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    with torch.cuda.device(dev):
        L.call("dbev_upsample2_zero_nhwc", L.ptr(g), L.ptr(gx), N, H, W, C, L.stream_ptr(dev))
    return gx


class _Conv1x1S2Bf6(Function):
    """F.conv2d(x, weight, stride=2) for a bias-free 1x1 filter (`eligible_s2`): the subsampled pixels are saved for the weight
    gradient instead of x (a quarter of the bytes)"""

    @staticmethod
    def forward(ctx, x, weight, stats=False):
        xs = subsample2(x)
        M = xs.shape[0] * xs.shape[2] * xs.shape[3]
        if ctx.needs_input_grad[0] and shape_ok(M, int(weight.shape[0]), int(weight.shape[1])):
            pack_both(weight, M)
        ctx.save_for_backward(xs, weight)
        ctx.hw = (int(x.shape[2]), int(x.shape[3]))
        if stats:
            y, part = product(xs, weight, stats=True)
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)              # no zero tensor for the (never used) gradient of the statistics output
            return y, part
        return product(xs, weight)

    @staticmethod
    def backward(ctx, gy, _gpart=None):
        xs, weight = ctx.saved_tensors
        if gy is None:
            return None, None, None
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = gw = None
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gxs = data_gradient(gy, weight) if need_x else None
        if need_w:
            gw = weight_gradient(xs, gy, weight)
        lib_x, lib_w = need_x and gxs is None, need_w and gw is None
        if lib_x or lib_w:                                    # the library's (stride-1) kernels on the subsampled pixels
            a, b, _ = torch.ops.aten.convolution_backward(gy, xs, weight, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                          [lib_x, lib_w, False])
            gxs = a.contiguous(memory_format=torch.channels_last) if lib_x else gxs
            gw = b if lib_w else gw
        if need_x:
            gx = upsample2_zero(gxs, *ctx.hw)
        return gx, gw, None


def conv1x1_s2(x, weight):
    """F.conv2d(x, weight, stride=2) for an `eligible_s2` pair, differentiable"""
    return _Conv1x1S2Bf6.apply(x, weight, False)


def conv1x1_s2_stats(x, weight):
    """-> (F.conv2d(x, weight, stride=2), partial statistics rows of the output)"""
    return _Conv1x1S2Bf6.apply(x, weight, True)


class _Conv1x1Bf6(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stats=False):
        M = x.shape[0] * x.shape[2] * x.shape[3]
        if ctx.needs_input_grad[0] and shape_ok(M, int(weight.shape[0]), int(weight.shape[1])):
            pack_both(weight, M)                              # the data gradient will want the transposed planes: one launch for both
        ctx.save_for_backward(x, weight)
        if stats:                                             # (bias-free layers in front of a training-mode norm)
            y, part = product(x, weight, stats=True)
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)              # no zero tensor for the (never used) gradient of the statistics output
            return y, part
        fused = _BIAS and bias is not None and bias.is_cuda and bias.dtype == torch.float32 and bias.is_contiguous()
        y = product(x, weight, bias=bias if fused else None)  # (round 6: the bias in the GEMM's epilogue, not a pass of its own)
        if bias is not None and not fused:
            y.add_(bias.view(1, -1, 1, 1))
        return y

    @staticmethod
    def backward(ctx, gy, _gpart=None):
        x, weight = ctx.saved_tensors
        if gy is None:
            return None, None, None, None
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = gw = gb = None
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if need_x:
            gx = data_gradient(gy, weight)
        if need_w:
            gw = weight_gradient(x, gy, weight)
        lib_x, lib_w = need_x and gx is None, need_w and gw is None
        if lib_x or lib_w:
            a, b, _ = torch.ops.aten.convolution_backward(gy, x, weight, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                          [lib_x, lib_w, False])
            gx = a if lib_x else gx
            gw = b if lib_w else gw
        if ctx.needs_input_grad[2]:
            from .colsum import channel_sum
            gb = channel_sum(gy)
        return gx, gw, gb, None


def conv1x1_stats(x, weight):
    """-> (F.conv2d(x, weight), partial statistics rows of the output) for an `eligible` pair: bn_act(..., pre=rows) skips its statistics pass"""
    return _Conv1x1Bf6.apply(x, weight, None, True)


def conv1x1(x, weight, bias=None):
    """F.conv2d(x, weight, bias) for an `eligible` pair (bias: added in a separate pass, its gradient by colsum.channel_sum)"""
    return _Conv1x1Bf6.apply(x, weight, bias)


# ---- 3x3 / stride-2 convolutions (conv2 of a stage-first bottleneck): implicit GEMM, forward only ---------------------------------------
_C3 = os.environ.get("DBEV_BF6_C3S2", "1") != "0"


def eligible_c3s2(x, weight, stride=(2, 2), padding=(1, 1), dilation=(1, 1), groups=1):
    """can `F.conv2d(x, weight, stride=2, padding=1)` (3x3) run as the implicit bf16x6 GEMM?  channels-last fp32 input AND filter (the
    filter's memory is the [Cout][9 Cin] matrix), the library's geometry rules (dbev_conv3x3s2_bf16x6_ok), a launch that fills the chip"""
    if not (_ON and _C3 and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and weight.dim() == 4):
        return False
    if tuple(weight.shape[2:]) != (3, 3) or tuple(stride) != (2, 2) or tuple(padding) != (1, 1) or tuple(dilation) != (1, 1) or groups != 1:
        return False
    N, C, H, W = x.shape
    Co = int(weight.shape[0])
    if not (weight.shape[1] == C and _nhwc(x) and _nhwc(weight) and L.lib().dbev_conv3x3s2_bf16x6_ok(N, H, W, C, Co)):
        return False
    M = N * (H // 2) * (W // 2)
    return -(-M // 128) * (Co // tile_n(M, Co)) >= _MIN_ITEMS


def product_c3s2(x, weight, stats=False):
    """conv2d(x, weight, stride 2, padding 1) of channels-last tensors (no autograd) [+ the output's partial statistics rows]"""
    dev = L.require_cuda(x, weight)
    N, C, H, W = x.shape
    Co = int(weight.shape[0])
    M = N * (H // 2) * (W // 2)
    tn = tile_n(M, Co)
    pack = packed(weight, False, tn)
    y = torch.empty((N, Co, H // 2, W // 2), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    part = torch.empty((int(L.call("dbev_gemm_bf16x6_stats_rows", M)), 2, Co), dtype=torch.float32, device=dev) if stats else None
    with torch.cuda.device(dev):
        L.call("dbev_conv3x3s2_bf16x6_forward_stats", L.ptr(x), L.ptr(pack), L.ptr(y), L.ptr(part), N, H, W, C, Co, int(tn), L.stream_ptr(dev))
    return (y, part) if stats else y


class _Conv3x3S2Bf6(Function):
    """forward on the implicit bf16x6 GEMM; both gradients are the library's (aten.convolution_backward) this round"""

    @staticmethod
    def forward(ctx, x, weight, stats=False):
        ctx.save_for_backward(x, weight)
        if stats:
            y, part = product_c3s2(x, weight, True)
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)
            return y, part
        return product_c3s2(x, weight)

    @staticmethod
    def backward(ctx, gy, _gpart=None):
        x, weight = ctx.saved_tensors
        if gy is None:
            return None, None, None
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx, gw, _ = torch.ops.aten.convolution_backward(gy, x, weight, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                                        [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
        return gx, gw, None


def conv3x3_s2(x, weight):
    return _Conv3x3S2Bf6.apply(x, weight, False)


def conv3x3_s2_stats(x, weight):
    return _Conv3x3S2Bf6.apply(x, weight, True)


class Bf6Conv3x3S2(nn.Conv2d):
    """nn.Conv2d(3x3, stride 2, padding 1, no bias) whose forward runs on the implicit bf16x6 GEMM when the input qualifies
    (`eligible_c3s2`); the stock convolution otherwise.  Same parameters and state-dict keys."""

    def forward(self, x):
        if self.bias is None and eligible_c3s2(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
            if not (torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)):
                return product_c3s2(x, self.weight)
            return conv3x3_s2(x, self.weight)
        return super().forward(x)


class Bf6Conv2d(nn.Conv2d):
    """nn.Conv2d (1x1, stride 1 or 2, no padding, no bias) whose forward and gradients run on the bf16x6 GEMM when the input qualifies
    (`eligible` / `eligible_s2`: stride 2 = subsample + GEMM); the stock convolution otherwise.  Same parameters and state-dict keys."""

    def forward(self, x):
        if self.bias is None and eligible(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
            if not (torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)):
                return product(x, self.weight)
            return conv1x1(x, self.weight)
        if self.bias is None and self.stride == (2, 2) and eligible_s2(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
            if not (torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)):
                return product(subsample2(x), self.weight)
            return conv1x1_s2(x, self.weight)
        return super().forward(x)


# ---- nn.Linear on [tokens, C] (round 6): the transformer's projections and FFNs of the BEVFormer recipe -----------------------------------
_LIN = os.environ.get("DBEV_BF6_LINEAR", "1") != "0"
_BIAS = os.environ.get("DBEV_BF6_BIAS", "1") != "0"          # the bias in the GEMM's epilogue (0: a separate pass, for A/B runs)


def _rows_as_nhwc(x2):
    """[M, K] row-major -> the same memory as a channels-last [1, K, M, 1] tensor (a view): the GEMM kernels' activation operand"""
    M, K = x2.shape
    return x2.view(1, M, 1, K).permute(0, 3, 1, 2)


def _rows_product(x2, weight, transposed, bias=None):
    """x2 [M, K] row-major times weight[N, K]^T (+ bias[N], in the kernel's epilogue; transposed: times weight [K', N'] itself, the data
    gradient) -> [M, N] row-major"""
    dev = L.require_cuda(x2, weight, bias)
    M, K = x2.shape
    cout = int(weight.shape[1 if transposed else 0])
    tn = tile_n(M, cout)
    pack = packed(weight, transposed, tn)
    y = torch.empty((M, cout), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        if bias is not None:
            L.call("dbev_gemm_bf16x6_forward_bias", L.ptr(x2), L.ptr(pack), L.ptr(bias), L.ptr(y), M, K, cout, K, int(tn), L.stream_ptr(dev))
        else:
            L.call("dbev_gemm_bf16x6_forward_stats", L.ptr(x2), L.ptr(pack), L.ptr(y), None, M, K, cout, K, int(tn), L.stream_ptr(dev))
    return y


def eligible_linear(x, weight):
    """can `F.linear(x, weight)` run on the bf16x6 GEMM?  fp32 device tensors, in / out features multiples of 64, enough rows to fill the
    chip (`shape_ok`: the decoder's 900 queries stay with the library, the encoder's 200 x 200 BEV queries come here)"""
    if not (_ON and _LIN and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.dim() == 2 and x.dim() >= 2):
        return False
    N, K = weight.shape
    if x.shape[-1] != K or not weight.is_contiguous():
        return False
    M = x.numel() // K
    return M * max(K, N) < 2 ** 31 and shape_ok(M, K, N)


class _LinearBf6(Function):
    """F.linear(x2, weight, bias) for x2 [M, K] contiguous: forward and both gradients on the bf16x6 kernels (the library where a shape
    rule says no), bias added in a separate pass / its gradient a column sum -- torch.nn.Linear's semantics (transformer_modules/*.py of
    the reference call plain nn.Linear)"""

    @staticmethod
    def forward(ctx, x2, weight, bias):
        M, K = x2.shape
        N = int(weight.shape[0])
        if ctx.needs_input_grad[0] and shape_ok(M, N, K):
            pack_both(weight, M)
        # [M, N] row-major, its own storage (autograd forbids returning a view); the bias in the kernel's epilogue
        fused = _BIAS and bias is not None and bias.is_contiguous() and bias.dtype == torch.float32
        y = _rows_product(x2, weight, False, bias if fused else None)
        if bias is not None and not fused:
            y.add_(bias)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x2, weight = ctx.saved_tensors
        gy = gy.contiguous()
        M, N = gy.shape
        gx = gw = gb = None
        g4 = _rows_as_nhwc(gy)
        if ctx.needs_input_grad[0]:
            K = int(weight.shape[1])
            gx = _rows_product(gy, weight, True) if shape_ok(M, N, K) else gy @ weight
        if ctx.needs_input_grad[1]:
            gw = weight_gradient(_rows_as_nhwc(x2), g4, weight)
            if gw is None:
                gw = gy.t() @ x2
        if ctx.has_bias and ctx.needs_input_grad[2]:
            from . import colsum
            gb = colsum.channel_sum(gy) if colsum.eligible(gy) else gy.sum(0)
        return gx, gw, gb


class Bf6Linear(nn.Linear):
    """nn.Linear whose product runs on the bf16x6 GEMM when `eligible_linear`; torch's otherwise.  Same parameters and state-dict keys."""

    def forward(self, x):
        if eligible_linear(x, self.weight):
            K = self.in_features
            x2 = x.reshape(-1, K)
            if not x2.is_contiguous():
                x2 = x2.contiguous()
            return _LinearBf6.apply(x2, self.weight, self.bias).view(*x.shape[:-1], self.out_features)
        return super().forward(x)


def use_bf6_linears(model):
    """Re-class the plain nn.Linear modules with in / out features multiples of 64; returns how many.  Idempotent."""
    if not (_ON and _LIN):
        return 0
    n = 0
    for m in model.modules():
        if type(m) is nn.Linear and m.in_features % 64 == 0 and m.out_features % 64 == 0:
            m.__class__ = Bf6Linear
            n += 1
    return n


def use_bf6_convs(model):
    """Re-class the bias-free 1x1 nn.Conv2d modules (stride 1, or stride 2: the `downsample` convolutions) with Cin % 64 == 0 and
    Cout % 64 == 0; returns how many.  Idempotent."""
    if not _ON:
        return 0
    n = 0
    for m in model.modules():
        if type(m) is nn.Conv2d and m.kernel_size == (1, 1) and m.stride in ((1, 1), (2, 2)) and m.padding == (0, 0) \
                and m.dilation == (1, 1) and m.groups == 1 and m.bias is None and m.in_channels % 64 == 0 and m.out_channels % 64 == 0:
            m.__class__ = Bf6Conv2d
            n += 1
        elif _C3 and type(m) is nn.Conv2d and m.kernel_size == (3, 3) and m.stride == (2, 2) and m.padding == (1, 1) \
                and m.dilation == (1, 1) and m.groups == 1 and m.bias is None and m.in_channels >= 64 \
                and (m.in_channels & (m.in_channels - 1)) == 0 and m.out_channels % 64 == 0:
            m.__class__ = Bf6Conv3x3S2
            n += 1
    return n
