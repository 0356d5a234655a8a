"""3x3 / stride 1 / pad 1 convolutions on the Winograd F(2x2, 3x3) fp32-MFMA kernels (csrc/wino.hip).

Host-side mirror of what the reference reaches through ``nn.Conv2d(k=3, s=1, p=1)`` -> cuDNN in its dense blocks
(mmdet3d/models/bricks/res_block.py:11-230, necks/lss_fpn.py:30-60, dense_heads/centerpoint_head.py:17-130,
backbones/second.py:60-78): same parameters, same state-dict keys, same autograd contract -- ``conv3x3(x, weight, bias)`` is
``F.conv2d(x, weight, bias, 1, 1)`` for channels-last fp32 device tensors with even H, W, Cin % 16 == 0, Cout % 64 == 0.
Forward and data gradient run the same kernel (the data gradient on grad_y with the rotated / transposed filters); the weight
gradient runs ``dbev_wino_conv3x3_backward_weight`` when the layer qualifies (Cin % 64 == 0) and the library's kernel otherwise.
"""
import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L

L.ensure_param_version_hook()          # fused optimizers do not move `_version`; the kept packs / coefficients follow it

# Work items (64-tile blocks x 64-channel blocks) a layer must offer before the Winograd kernels take it -- measured
# (tools/kbench_wino.py, profiles/r04_wino_vs_miopen.txt): with the two-workgroups-per-CU kernel for small grids, 64 items run at 0.9 x
# the library's kernel, 128 at 1.7 x, 384 at 1.3 x, larger layers at 1.6-2.2 x.
_MIN_WG = int(os.environ.get("DBEV_WINO_MIN_WG", "100"))
_WGRAD = os.environ.get("DBEV_WINO_WGRAD", "1") != "0"


def _nhwc(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def eligible(x, weight, stride=(1, 1), padding=(1, 1), dilation=(1, 1), groups=1):
    """can `F.conv2d(x, weight, ...)` run on the Winograd kernels?"""
    if os.environ.get("DBEV_WINO", "1") == "0":
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and weight.dim() == 4):
        return False
    if tuple(weight.shape[2:]) != (3, 3) or tuple(stride) != (1, 1) or tuple(padding) != (1, 1) or tuple(dilation) != (1, 1) or groups != 1:
        return False
    N, C, H, W = x.shape
    Co = weight.shape[0]
    if weight.shape[1] != C or H % 2 or W % 2 or C % 4 or Co % 64 or not _nhwc(x) or H * W * max(C, Co) * 4 >= 2 ** 31:
        return False
    return N * H * W * max(C, Co) < 2 ** 31 - 1


def worthwhile(x, Cout):
    """does the layer offer enough workgroups for the kernel to beat the library's (see _MIN_WG)?"""
    N, _, H, W = x.shape
    return _blocks(N, H, W) * (Cout // 64) >= _MIN_WG


def _blocks(N, H, W):
    """tile blocks of the kernel's plan (wino_plan in csrc/wino.hip): 8 x 8 or 4 x 16 tiles of 2 x 2 outputs, whichever wastes less"""
    th, tw = H // 2, W // 2
    return N * min(-(-th // 8) * -(-tw // 8), -(-th // 4) * -(-tw // 16))


def pack_filters(weight, data_gradient=False, for_input=None, out=None):
    """G g G^T of every (co, c) filter in the kernels' consumption orders (dbev_wino_filter_pack).  `for_input`: shape [N, K, H, W] of
    the tensor the packed filters will be applied to -- only the format of the forward kernel that layer gets is written (the other
    slot of the buffer stays uninitialised); None: both formats.  out: a buffer of an earlier call for the same layer, written again
    (a captured hipGraph may be reading through its address: graphed.py)."""
    dev = L.require_cuda(weight)
    Co, C = weight.shape[:2]
    K, J = (Co, C) if data_gradient else (C, Co)
    flags = int(bool(data_gradient))
    if for_input is not None:
        N, _, H, W = for_input
        ver = int(L.call("dbev_wino_conv3x3_forward_kernel", N, H, W, K, J))
        flags |= {3: 2, 2: 4}.get(ver, 0)
    n = int(L.call("dbev_wino_filter_floats", K, J))
    if n == 0:
        raise L.DbevHipError(f"wino: unsupported channel counts {C} -> {Co} (data_gradient={data_gradient})")
    packed = out if (out is not None and out.numel() == n and out.device == dev) else torch.empty((n,), dtype=torch.float32, device=dev)
    so, sc, sa, sb = weight.stride()
    with torch.cuda.device(dev):
        L.call("dbev_wino_filter_pack", L.ptr(weight), so, sc, sa, sb, Co, C, flags, L.ptr(packed), L.stream_ptr(dev))
    return packed


def packed_pair(weight, x_shape, want_dgrad):
    """-> (forward pack, data-gradient pack or None) of `weight` for inputs of shape `x_shape` [N, C, H, W]: both directions come out
    of ONE launch (dbev_wino_filter_pack_pair), each in the format of the kernel its direction gets, and stay attached to the weight
    until it changes (version / storage) -- a layer's forward in the detached frame, its forward in the differentiated frame and its
    data gradient share one pack per step."""
    dev = L.require_cuda(weight)
    N, C, H, W = (int(v) for v in x_shape)
    Co = int(weight.shape[0])
    want_dgrad = bool(want_dgrad) and C % 64 == 0
    key = (weight._version, weight.data_ptr(), N, H, W)
    hit = getattr(weight, "_dbev_wino_pair", None)
    fwd = dgrad = None
    if hit is not None and hit[0] == key:
        L.check_fingerprint(hit[3], "Winograd filter pack", weight)
        fwd, dgrad = hit[1], hit[2]
        if dgrad is not None or not want_dgrad:
            L.note_derived("wino_pair", weight, ((N, C, H, W), want_dgrad), (fwd, dgrad))
            return fwd, dgrad
    # a STALE entry of the same storage and input size (the weight's version moved: an optimizer step, load_state_dict): its buffers
    # are written again rather than replaced -- their addresses stay what a captured hipGraph (graphed.py) baked in
    old_f = old_d = None
    if hit is not None and hit[0] != key and hit[0][1:] == key[1:] and hit[1].device == dev:
        old_f, old_d = hit[1], hit[2]
        want_dgrad = want_dgrad or old_d is not None         # (a kept data-gradient pack is refreshed with its forward pack)
    fk = 0 if fwd is not None else int(L.call("dbev_wino_conv3x3_forward_kernel", N, H, W, C, Co))
    dk = int(L.call("dbev_wino_conv3x3_forward_kernel", N, H, W, Co, C)) if want_dgrad else 0
    if fwd is None:
        if fk == 0:
            raise L.DbevHipError(f"wino: unsupported layer {C} -> {Co} at {H} x {W}")
        fwd = old_f if old_f is not None else torch.empty((int(L.call("dbev_wino_filter_floats", C, Co)),), dtype=torch.float32, device=dev)
    if dk:
        dgrad = old_d if old_d is not None else torch.empty((int(L.call("dbev_wino_filter_floats", Co, C)),), dtype=torch.float32, device=dev)
    so, sc, sa, sb = weight.stride()
    with torch.cuda.device(dev):
        L.call("dbev_wino_filter_pack_pair", L.ptr(weight), so, sc, sa, sb, Co, C, fk, dk, L.ptr(fwd), L.ptr(dgrad), L.stream_ptr(dev))
    try:
        weight._dbev_wino_pair = (key, fwd, dgrad, L.fingerprint(weight))
    except AttributeError:                                  # a tensor type without instance attributes: no reuse, still correct
        pass
    L.note_derived("wino_pair", weight, ((N, C, H, W), want_dgrad), (fwd, dgrad))
    return fwd, dgrad


def stats_rows(x_shape, Cout):
    N, C, H, W = x_shape
    return int(L.call("dbev_wino_conv3x3_stats_rows", N, H, W, C, Cout))


COUNTERS = {"on": False, "launches": 0, "flops": 0, "bytes": 0}      # bench.py: algorithmic work of the forward / data-gradient launches


def conv_packed(x, packed, Cout, bias=None, stats=False, relu=False):
    """one launch of the convolution kernel: x [N, C, H, W] channels-last, packed filters for C -> Cout.
    -> y (channels-last) or (y, partial statistics rows f32[rows, 2, Cout]) with stats=True"""
    dev = L.require_cuda(x, packed, bias)
    N, C, H, W = x.shape
    if COUNTERS["on"]:
        COUNTERS["launches"] += 1
        COUNTERS["flops"] += 32 * N * (H // 2) * (W // 2) * C * Cout              # Winograd-domain products (direct: x 2.25)
        COUNTERS["bytes"] += 4 * (N * H * W * (C + Cout) + 16 * C * Cout)          # x read once, y written once, packed filters
    y = torch.empty((N, Cout, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    part = torch.empty((stats_rows(x.shape, Cout), 2, Cout), dtype=torch.float32, device=dev) if stats else None
    with torch.cuda.device(dev):
        L.call("dbev_wino_conv3x3_forward_act", L.ptr(x), L.ptr(packed), L.ptr(bias), L.ptr(y), L.ptr(part), N, H, W, C, Cout,
               int(bool(relu)), L.stream_ptr(dev))
    return (y, part) if stats else y


# ---- convolution -> eval-mode norm -> ReLU of a frozen stack in one launch -----------------------------------------------------------
def fold_ready(conv, norm, x):
    """may `conv -> norm(+ReLU)` run as ONE Winograd launch with the norm folded in?  A re-classed 3x3 convolution followed by a fused
    BatchNorm + ReLU module in EVAL mode (running statistics), nothing differentiable involved (the frozen teacher: second.py:60-78,
    centerpoint_head.py:17-130), the input fit for the kernels."""
    from . import bn_act as BA
    if not (type(conv) is WinoConv2d and type(norm) is BA.BatchNormAct2d and BA._state["enabled"]):
        return False
    if norm.training or norm.running_mean is None or not norm.affine or norm.num_features != conv.out_channels:
        return False
    if torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad or norm.weight.requires_grad
                                    or (conv.bias is not None and conv.bias.requires_grad)):
        return False
    return (conv.padding_mode == "zeros" and eligible(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
            and worthwhile(x, conv.out_channels))


def folded_pack(conv, norm, x_shape, dev):
    """-> (key, norm coefficient key, packed filters with the norm's scale folded in, bias = the norm's shift (+ the scaled convolution
    bias), fingerprint) of a `fold_ready` pair, kept on the convolution until its or the norm's tensors change; a stale entry is
    re-derived INTO ITS BUFFERS (stable addresses for a captured hipGraph: graphed.py)"""
    from . import bn_act as BA
    coef = BA._eval_coef(norm, dev)                          # scale | shift
    ckey = norm.__dict__["_dbev_eval_coef"][0]               # changes whenever one of the norm's four tensors does
    w = conv.weight
    N, _, H, W = (int(v) for v in x_shape)
    key = (w._version, w.data_ptr(), None if conv.bias is None else conv.bias._version, N, H, W)
    hit = conv.__dict__.get("_dbev_wino_folded")
    if hit is None or hit[0] != key or hit[1] != ckey:
        Co = conv.out_channels
        reuse = hit is not None and hit[0][1] == key[1] and hit[0][3:] == key[3:] and hit[2].device == torch.device(dev)
        with torch.no_grad():
            scale, shift = coef[:Co], coef[Co:]
            wf = w.detach() * scale.view(Co, 1, 1, 1)
            b = (shift if conv.bias is None else shift + conv.bias.detach() * scale).contiguous()
            U = pack_filters(wf, False, x_shape, out=hit[2] if reuse else None)
            if reuse:
                b = hit[3].copy_(b)
            elif b.data_ptr() == coef.data_ptr() or b._base is not None:
                b = b.clone()                                # own storage: the coefficient tensor is rewritten in place when the norm moves
        hit = (key, ckey, U, b, L.fingerprint(w, conv.bias))
        conv.__dict__["_dbev_wino_folded"] = hit
    else:
        L.check_fingerprint(hit[4], "folded convolution + norm filter pack", w, conv.bias)
    L.note_derived("wino_folded", conv, (norm, (N, int(x_shape[1]), H, W), dev), (hit[2], hit[3]))
    return hit


def conv_norm_relu_eval(x, conv, norm):
    """relu(norm(conv(x))) for a `fold_ready` triple: the norm's scale goes into the filters before they are transformed and packed
    (kept until the convolution's or the norm's tensors change), its shift (+ the scaled convolution bias) is the kernel's bias, the
    ReLU its output flag -- no normalisation pass over the output."""
    hit = folded_pack(conv, norm, x.shape, x.device)
    return conv_packed(x, hit[2], conv.out_channels, hit[3], relu=True)


class ConvNormSequential(nn.Sequential):
    """nn.Sequential whose `3x3 convolution -> eval-mode norm + ReLU` neighbours run as one launch when `fold_ready`; every other
    module, and every pair that is not ready (training mode, gradients), is called as nn.Sequential calls it.  Same children, same
    state-dict keys (link_conv_norm_stacks re-classes in place)."""

    def forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            if i + 1 < len(mods) and torch.is_tensor(x) and fold_ready(mods[i], mods[i + 1], x):
                x = conv_norm_relu_eval(x, mods[i], mods[i + 1])
                i += 2
            else:
                x = mods[i](x)
                i += 1
        return x


def link_conv_norm_stacks(model):
    """Re-class the plain nn.Sequential stacks that hold a (WinoConv2d, fused BatchNorm + ReLU) neighbour pair; returns how many pairs.
    Run after bn_act.fuse_bn_relu_modules and use_wino_convs.  Idempotent."""
    from . import bn_act as BA
    n = 0
    for mod in model.modules():
        if type(mod) in (nn.Sequential, ConvNormSequential):
            kids = list(mod._modules.values())
            pairs = sum(1 for a, b in zip(kids, kids[1:]) if type(a) is WinoConv2d and type(b) is BA.BatchNormAct2d)
            if pairs:
                mod.__class__ = ConvNormSequential
                n += pairs
    return n


class _Conv3x3Wino(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stats):
        Co = weight.shape[0]
        U, ctx.dgrad_pack = packed_pair(weight, x.shape, ctx.needs_input_grad[0])
        ctx.pack_key = (weight._version, weight.data_ptr())
        out = conv_packed(x, U, Co, bias, stats)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if stats:
            ctx.mark_non_differentiable(out[1])
            ctx.set_materialize_grads(False)              # no zero tensor for the (never used) gradient of the statistics output
        return out

    @staticmethod
    def backward(ctx, gy, _gpart=None):
        x, weight = ctx.saved_tensors
        if gy is None:
            return None, None, None, None
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            C = weight.shape[1]
            if C % 64 == 0:
                Ud = ctx.dgrad_pack if ctx.pack_key == (weight._version, weight.data_ptr()) else None
                gx = conv_packed(gy, Ud if Ud is not None else pack_filters(weight, True, gy.shape), C)
            else:                                          # e.g. a 16- or 32-channel input: the library's data gradient
                gx = torch.ops.aten.convolution_backward(gy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            gw = weight_gradient(x, gy, weight)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            from .colsum import channel_sum
            gb = channel_sum(gy)                           # one streaming pass, fixed order (csrc/colsum.hip)
        return gx, gw, gb, None


def weight_gradient(x, gy, weight):
    """grad of the 3x3 filters: Winograd-domain GEMMs over the tiles (dbev_wino_conv3x3_backward_weight, fixed summation order) when
    both channel counts are multiples of 64, the library's kernel otherwise; returned with `weight`'s strides"""
    N, C, H, W = x.shape
    Co = weight.shape[0]
    nbytes = int(L.call("dbev_wino_conv3x3_backward_weight_workspace_bytes", N, H, W, C, Co)) if _WGRAD else 0
    if nbytes == 0 or not (_nhwc(x) and _nhwc(gy)):
        return torch.ops.aten.convolution_backward(gy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    dev = x.device
    gw = torch.empty_like(weight)                       # preserve_format: the parameter's own strides
    so, sc, sa, sb = gw.stride()
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.call("dbev_wino_conv3x3_backward_weight", L.ptr(x), L.ptr(gy), L.ptr(gw), so, sc, sa, sb, N, H, W, C, Co, L.ptr(ws), nbytes,
               L.stream_ptr(dev))
    return gw


def conv3x3(x, weight, bias=None):
    """F.conv2d(x, weight, bias, stride=1, padding=1) for an `eligible` pair, differentiable"""
    return _Conv3x3Wino.apply(x, weight, bias, False)


def conv3x3_stats(x, weight, bias=None):
    """-> (conv, partial rows of (sum y, sum y^2) per channel) for the fused BatchNorm that follows (bn_act(..., pre=rows))"""
    return _Conv3x3Wino.apply(x, weight, bias, True)


class WinoConv2d(nn.Conv2d):
    """nn.Conv2d (3x3, stride 1, pad 1) whose forward and data gradient run on the Winograd kernels when the input qualifies
    (`eligible`); the stock convolution otherwise.  Same parameters and state-dict keys (use_wino_convs re-classes in place).  With
    gradients disabled (the frozen teacher, the student's detached frame) the packed filters are kept until the weight changes."""

    def forward(self, x):
        if self.padding_mode == "zeros" and eligible(x, self.weight, self.stride, self.padding, self.dilation, self.groups) \
                and worthwhile(x, self.out_channels):
            if not (torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)):
                return conv_packed(x, packed_pair(self.weight, x.shape, self.weight.requires_grad)[0], self.out_channels, self.bias)
            return conv3x3(x, self.weight, self.bias)
        return super().forward(x)


def _wino_geometry(m):
    return (m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) and m.dilation == (1, 1) and m.groups == 1
            and m.padding_mode == "zeros" and m.in_channels % 4 == 0 and m.out_channels % 64 == 0)


def use_wino_convs(model):
    """Re-class the nn.Conv2d modules with the Winograd geometry (3x3, s1, p1, Cin % 16 == 0, Cout % 64 == 0); returns how many.
    Parameters and state-dict keys are untouched.  Idempotent."""
    if os.environ.get("DBEV_WINO", "1") == "0":
        return 0
    n = 0
    for m in model.modules():
        if type(m) is nn.Conv2d and _wino_geometry(m):
            m.__class__ = WinoConv2d
            n += 1
    return n


def conv3x3_bn_ready(conv, bn, x):
    """conv -> training-mode BatchNorm2d pairs whose batch statistics the convolution's epilogue can take (bn_act(..., pre=rows))"""
    from . import bn_act as BA
    if not (type(conv) is WinoConv2d and BA._state["enabled"] and conv.bias is None):       # (also without autograd: the detached frame)
        return False
    if not (type(bn) in BA._BN_TYPES and bn.affine and bn.training and bn.momentum is not None and bn.running_mean is not None
            and BA._channels_ok(conv.out_channels)):
        return False
    return eligible(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups) and worthwhile(x, conv.out_channels)
