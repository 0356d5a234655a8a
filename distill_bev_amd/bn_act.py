"""BatchNorm2d fused with the residual add and ReLU that follow it, on the gfx950 kernels of csrc/bn_act.hip.

The reference chains `norm -> (+ identity) -> relu` as separate modules (mmdet ResNet Bottleneck, mmdet3d
bricks/res_block.py:11-100, mmcv ConvModule, the nn.Sequential stacks of second.py:60-78 / lss_fpn.py:30-60);
on MI355X those are 20 % of the training step in HBM-bound full-tensor passes.  `bn_act()` computes exactly
`relu(batch_norm(x) + residual)` (training: batch statistics + running-stat update; eval under no_grad: running
statistics) in 3 passes forward / 5 backward instead of 5-7 / 8.  `fuse_bn_relu_modules()` rewires a built model
WITHOUT touching parameter names: a BatchNorm2d followed by nn.ReLU becomes a `BatchNormAct2d` (same
parameters/buffers, same state-dict keys) and the ReLU becomes nn.Identity.

Anything the kernels do not cover (CPU tensors, NCHW-contiguous activations, channel counts that are not
4 * 2^k, eval mode with autograd, SyncBN, momentum=None) takes the unfused torch ops -- still the GPU path of
the reference's own op sequence, not a CPU fallback.
"""
import contextlib
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib as L

L.ensure_param_version_hook()          # fused optimizers do not move `_version`; the kept packs / coefficients follow it

_state = {"enabled": os.environ.get("DBEV_FUSED_BN", "1") != "0",
          "fork": os.environ.get("DBEV_BN_FORK", "1") != "0",      # residual-block outputs carry a second handle (see _BNActTrain.forward)
          # round 5: a residual norm's forward writes its ReLU gate as one byte per four channels and the two backward passes read
          # that instead of the saved output (16 x fewer bytes on the largest tensors of the step); 0: read the output as before
          "gate_mask": os.environ.get("DBEV_BN_GATE_MASK", "1") != "0"}


@contextlib.contextmanager
def disabled():
    """Run the enclosed code with the unfused torch op sequence (used by the parity tests)."""
    old = _state["enabled"]
    _state["enabled"] = False
    try:
        yield
    finally:
        _state["enabled"] = old


def _channels_ok(C):
    if C % 4:
        return False
    c4 = C // 4
    return (c4 & (c4 - 1)) == 0 if c4 <= 256 else c4 % 256 == 0


def _nhwc(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def eligible(x, bn, residual=None):
    if not (_state["enabled"] and x.is_cuda and x.dtype == torch.float32 and _nhwc(x) and type(bn) in _BN_TYPES
            and bn.affine and _channels_ok(x.shape[1]) and x.numel() > 0):
        return False
    if residual is not None and not (residual.shape == x.shape and residual.dtype == x.dtype and _nhwc(residual)):
        return False
    use_batch_stats = bn.training or bn.running_mean is None
    if use_batch_stats:
        if x.numel() // x.shape[1] <= 1:      # torch raises "Expected more than 1 value per channel": keep that
            return False
        return bn.momentum is not None or bn.running_mean is None
    return not (torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad
                                             or (residual is not None and residual.requires_grad)))


def _why_not(x, bn, residual):
    """one-line reason for the fallback ledger"""
    if x.dtype != torch.float32:
        return f"dtype {x.dtype}"
    if not _nhwc(x) or (residual is not None and not _nhwc(residual)):
        return "not channels-last"
    if type(bn) not in _BN_TYPES or not bn.affine:
        return f"norm type {type(bn).__name__}"
    if not _channels_ok(x.shape[1]):
        return f"{x.shape[1]} channels"
    if bn.training or bn.running_mean is None:
        return "one value per channel" if x.numel() // x.shape[1] <= 1 else "momentum=None"
    return "eval-mode norm inside autograd"


def _alias(y):
    """a second tensor on y's storage with no view relation to it (a plain `y.view(...)` returned from a Function is tracked as a view
    of the first output)"""
    return torch.empty((0,), dtype=y.dtype, device=y.device).set_(y.untyped_storage(), y.storage_offset(), y.size(), y.stride())


def _two_addends(dy, dy2):
    """incoming gradients of a forked output -> (first, second or None), channels-last"""
    if dy is None:
        dy, dy2 = dy2, None
    cl = lambda t: None if t is None else t.contiguous(memory_format=torch.channels_last)
    return cl(dy), cl(dy2)


def forked(t):
    """the second handle of a forked block output (bn_act(..., fork=True)), or t itself"""
    return getattr(t, "_dbev_fork", t)


class _BNActTrain(Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, nbt, momentum, eps, relu, pre=None, fork=False):
        """pre: f32[rows, 2, C] partial (sum, sum of squares) rows the producer of x already took (conv1x1_stats): no statistics pass.
        fork: return the output TWICE (two tensors on one storage): the next residual block hands one to its first convolution and
        the other to its identity branch, their gradients come back separately and the backward kernels add them on the way in --
        autograd's own accumulation is a full read-read-write pass per junction (2.6 ms of the step)."""
        dev = x.device
        N, C, H, W = x.shape
        M = N * H * W
        y = torch.empty_like(x)
        save_mean = torch.empty((C,), dtype=torch.float32, device=dev)
        save_invstd = torch.empty((C,), dtype=torch.float32, device=dev)
        coef = torch.empty((2 * C,), dtype=torch.float32, device=dev)
        nbytes = L.lib().dbev_bn_act_workspace_bytes(M, C)
        ws = torch.empty((nbytes + 12 * C,), dtype=torch.uint8, device=dev)
        need_y = relu and residual is not None
        # the backward of a residual norm needs the output only for the ReLU gate: one byte per four channels instead
        gmask = (torch.empty((M * C // 4,), dtype=torch.uint8, device=dev)
                 if need_y and _state["gate_mask"] and any(ctx.needs_input_grad[:4]) else None)
        with torch.cuda.device(dev):
            L.call("dbev_bn_act_train_forward_mask", L.ptr(x), L.ptr(residual), L.ptr(weight), L.ptr(bias),
                   L.ptr(running_mean), L.ptr(running_var), L.ptr(nbt), float(momentum or 0.0), float(eps), int(relu), L.ptr(y),
                   L.ptr(save_mean), L.ptr(save_invstd), L.ptr(coef), M, C, L.ptr(pre), 0 if pre is None else pre.shape[0],
                   L.ptr(gmask), L.ptr(ws), ws.numel(), L.stream_ptr(dev),
                   alg_bytes=4 * M * C * (3 + (residual is not None) - (pre is not None)))     # x (stats), x (apply) [+ res] + y
        L.touched(running_mean, running_var, nbt)
        y2 = _alias(y) if fork else None
        # both handles of a forked output are SAVED (whether or not the backward reads y): they share one storage but have separate
        # version counters, so an in-place op on either handle (`feat += ...`, `relu_`) would silently change what the other handle's
        # consumer saved for its backward -- saved, autograd's version check turns that into its usual "modified by an inplace
        # operation" error when this node's backward unpacks them.  Contract: forked outputs are never written in place.
        # (the gate mask travels through save_for_backward like everything else the backward reads: saved-tensor hooks / offloading see
        # it -- ADVICE r5; when it exists the backward does not read y, which stays saved only as a forked handle, see above)
        ctx.save_for_backward(x, y if ((need_y and gmask is None) or fork) else None, weight, save_mean, save_invstd, coef, y2, gmask)
        ctx.cfg = (M, C, bool(relu), residual is not None, need_y)
        if fork:
            ctx.set_materialize_grads(False)
            return y, y2
        return y

    @staticmethod
    def backward(ctx, dy, dy2=None):
        x, y, weight, save_mean, save_invstd, coef, _y2, gmask = ctx.saved_tensors      # unpacking checks the versions of both handles
        M, C, relu, has_res, need_y = ctx.cfg
        if not need_y:
            y = None
        dy, dy2 = _two_addends(dy, dy2)
        if dy is None:
            return (None,) * 12
        dev = dy.device
        dx = torch.empty_like(x)
        # without ReLU the residual branch receives dy itself: nothing to write
        dres = torch.empty_like(x) if (has_res and relu) else None
        dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
        nbytes = L.lib().dbev_bn_act_workspace_bytes(M, C)
        ws = torch.empty((nbytes + 12 * C,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            gm = gmask if need_y else None
            L.call("dbev_bn_act_backward3", L.ptr(dy), L.ptr(dy2), L.ptr(x), L.ptr(y if gm is None else gm), int(gm is not None),
                   L.ptr(weight), L.ptr(save_mean), L.ptr(save_invstd), L.ptr(coef), int(relu), L.ptr(dx), L.ptr(dres), L.ptr(dgamma),
                   L.ptr(dbeta), M, C, L.ptr(ws), ws.numel(), L.stream_ptr(dev),
                   alg_bytes=4 * M * C * (5 + 3 * (dres is not None) + 2 * (dy2 is not None)))   # dy [+ dy2], x twice each + dx [+ y twice + dres]
        if has_res and not relu:                    # without ReLU the residual branch receives the incoming gradient itself
            dres = dy if dy2 is None else dy + dy2
        return dx, dres if has_res else None, dgamma, dbeta, None, None, None, None, None, None, None, None


class _BNDualTrain(Function):
    """relu(bn(x) + bn_d(xd)): dbev_bn_dual_train_forward / dbev_bn_dual_backward"""

    @staticmethod
    def forward(ctx, x, xd, w, b, rm, rv, nbt, mom, eps, wd, bd, rmd, rvd, nbtd, momd, epsd, relu, pre=None, pre_d=None, fork=False):
        dev = x.device
        N, C, H, W = x.shape
        M = N * H * W
        y = torch.empty_like(x)
        stats = torch.empty((8, C), dtype=torch.float32, device=dev)      # mean, invstd, scale, shift of both norms
        nbytes = L.lib().dbev_bn_dual_workspace_bytes(M, C)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        gmask = (torch.empty((M * C // 4,), dtype=torch.uint8, device=dev)
                 if relu and _state["gate_mask"] and any(ctx.needs_input_grad[:4]) else None)     # see _BNActTrain.forward
        with torch.cuda.device(dev):
            L.call("dbev_bn_dual_train_forward_mask", L.ptr(x), L.ptr(xd), L.ptr(w), L.ptr(b), L.ptr(rm), L.ptr(rv), L.ptr(nbt),
                   float(mom or 0.0), float(eps), L.ptr(wd), L.ptr(bd), L.ptr(rmd), L.ptr(rvd), L.ptr(nbtd), float(momd or 0.0),
                   float(epsd), int(relu), L.ptr(y), L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(stats[2:4]), L.ptr(stats[4]),
                   L.ptr(stats[5]), L.ptr(stats[6:8]), M, C, L.ptr(pre), 0 if pre is None else pre.shape[0], L.ptr(pre_d),
                   0 if pre_d is None else pre_d.shape[0], L.ptr(gmask), L.ptr(ws), ws.numel(), L.stream_ptr(dev),
                   alg_bytes=4 * M * C * (5 - (pre is not None) - (pre_d is not None)))
        L.touched(rm, rv, nbt, rmd, rvd, nbtd)
        y2 = _alias(y) if fork else None
        ctx.save_for_backward(x, xd, y if ((relu and gmask is None) or fork) else None, w, wd, stats, y2, gmask)     # both handles saved: see _BNActTrain.forward
        ctx.cfg = (M, C, bool(relu))
        if fork:
            ctx.set_materialize_grads(False)
            return y, y2
        return y

    @staticmethod
    def backward(ctx, dy, dy2=None):
        x, xd, y, w, wd, stats, _y2, gmask = ctx.saved_tensors
        M, C, relu = ctx.cfg
        if not relu:
            y = None
        dy, dy2 = _two_addends(dy, dy2)
        if dy is None:
            return (None,) * 20
        dev = dy.device
        dx, dxd = torch.empty_like(x), torch.empty_like(xd)
        g = torch.empty((4, C), dtype=torch.float32, device=dev)
        nbytes = L.lib().dbev_bn_dual_workspace_bytes(M, C)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            gm = gmask if relu else None
            L.call("dbev_bn_dual_backward3", L.ptr(dy), L.ptr(dy2), L.ptr(x), L.ptr(xd), L.ptr(y if gm is None else gm), int(gm is not None),
                   L.ptr(w), L.ptr(stats[0]),
                   L.ptr(stats[1]), L.ptr(wd), L.ptr(stats[4]), L.ptr(stats[5]), int(relu), L.ptr(dx), L.ptr(dxd), L.ptr(g[0]),
                   L.ptr(g[1]), L.ptr(g[2]), L.ptr(g[3]), M, C, L.ptr(ws), ws.numel(), L.stream_ptr(dev),
                   alg_bytes=4 * M * C * (10 + 2 * (dy2 is not None)))
        return (dx, dxd, g[0], g[1], None, None, None, None, None, g[2], g[3], None, None, None, None, None, None, None, None, None)


def _fork_out(out, fork):
    if fork and _state.get("fork", True):
        y, y2 = out
        y._dbev_fork = y2
        return y
    return out


def bn_act_dual(x, bn, xd, bn_d, relu=True, pre=None, pre_d=None, fork=False):
    """relu(bn(x) + bn_d(xd)) -- the tail of a residual block whose identity branch ends in its own BatchNorm (`downsample`).
    Training mode on channels-last tensors: one fused forward / backward that never writes bn_d(xd) or the gated gradient;
    otherwise the same value through bn_act(x, bn, residual=bn_d(xd))."""
    if (eligible(x, bn, xd) and eligible(xd, bn_d) and bn.training and bn_d.training and bn.running_mean is not None
            and bn_d.running_mean is not None and torch.is_grad_enabled()):
        fk = bool(fork and _state.get("fork", True))
        return _fork_out(_BNDualTrain.apply(x, xd, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                            bn.momentum, bn.eps, bn_d.weight, bn_d.bias, bn_d.running_mean, bn_d.running_var,
                                            bn_d.num_batches_tracked, bn_d.momentum, bn_d.eps, relu, pre, pre_d, fk), fk)
    return bn_act(x, bn, bn_act(xd, bn_d, None, False, pre=pre_d), relu, pre=pre, fork=fork)


class _Conv1x1Stats(Function):
    """z = conv1x1(x, w) on the fp32 matrix cores with the BatchNorm partial statistics of z in the epilogue
    (dbev_conv1x1_forward); backward = the library convolution's data / weight gradients."""

    @staticmethod
    def forward(ctx, x, w):
        dev = x.device
        N, Ci, H, W = x.shape
        Co = w.shape[0]
        M = N * H * W
        z = torch.empty((N, Co, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        rows = int(L.call("dbev_conv1x1_stats_rows", M, Ci, Co))
        part = torch.empty((rows, 2, Co), dtype=torch.float32, device=dev)
        w2 = w.reshape(Co, Ci)
        w2 = w2 if w2.is_contiguous() else w2.contiguous()
        with torch.cuda.device(dev):
            L.call("dbev_conv1x1_forward", L.ptr(x), L.ptr(w2), L.ptr(z), L.ptr(part), M, Ci, Co, Ci, L.stream_ptr(dev),
                   alg_bytes=4 * M * (Ci + Co))
        ctx.save_for_backward(x, w)
        ctx.mark_non_differentiable(part)
        ctx.set_materialize_grads(False)                  # no zero tensor for the (never used) gradient of the statistics output
        return z, part

    @staticmethod
    def backward(ctx, gz, _gpart):
        x, w = ctx.saved_tensors
        if gz is None:
            return None, None
        gz = gz.contiguous(memory_format=torch.channels_last)
        from .gemm_bf6 import data_gradient, weight_gradient
        gx = data_gradient(gz, w) if ctx.needs_input_grad[0] else None      # bf16x6 GEMMs where they apply, else the library
        gw = weight_gradient(x, gz, w) if ctx.needs_input_grad[1] else None
        lib_x, lib_w = ctx.needs_input_grad[0] and gx is None, ctx.needs_input_grad[1] and gw is None
        if lib_x or lib_w:
            a, b, _ = torch.ops.aten.convolution_backward(gz, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [lib_x, lib_w, False])
            gx = a if lib_x else gx
            gw = b if lib_w else gw
        return gx, gw


_C1 = {"enabled": os.environ.get("DBEV_CONV1X1", "1") != "0", "max_cin": int(os.environ.get("DBEV_CONV1X1_MAX_CIN", "512")),
       "min_rows": int(os.environ.get("DBEV_CONV1X1_MIN_ROWS", "300000"))}


def conv1x1_bn_ready(conv, bn, x):
    """The 1x1 convolution + training-mode BatchNorm pairs the fused GEMM serves: where `conv + statistics pass` is slower in the
    library than the hand-written kernel with the statistics in its epilogue -- measured (tools/kbench_c1x1.py,
    profiles/r03_conv1x1_vs_miopen.txt): the >= 300 k-pixel maps of the image backbone's first stage (64 <-> 256 channels, HBM-bound:
    conv + statistics 0.21 vs 0.30 ms); at stage 2 the two are even, from stage 3 on the library's kernels win."""
    if not (_C1["enabled"] and _state["enabled"] and type(conv).__name__ in ("Conv2d", "Bf6Conv2d") and isinstance(conv, nn.Conv2d)
            and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None):
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and _nhwc(x) and conv.in_channels % 32 == 0 and conv.out_channels % 32 == 0
            and conv.in_channels <= _C1["max_cin"] and x.shape[0] * x.shape[2] * x.shape[3] >= _C1["min_rows"]):
        return False
    if not (type(bn) in _BN_TYPES and bn.affine and bn.training and bn.momentum is not None and bn.running_mean is not None
            and _channels_ok(conv.out_channels)):
        return False
    return int(L.call("dbev_conv1x1_stats_rows", x.shape[0] * x.shape[2] * x.shape[3], conv.in_channels, conv.out_channels)) > 0


def conv1x1_stats(conv, x):
    """-> (conv(x), partial statistics rows) for a pair accepted by conv1x1_bn_ready"""
    return _Conv1x1Stats.apply(x, conv.weight)


def split_downsample(downsample):
    """(conv, norm) if `downsample` is the conv -> BatchNorm2d pair mmdet's ResLayer builds, else None"""
    if isinstance(downsample, nn.Sequential) and len(downsample) == 2 and isinstance(downsample[0], nn.Conv2d) \
            and type(downsample[1]) in _BN_TYPES and not isinstance(downsample[1], BatchNormAct2d):
        return downsample[0], downsample[1]
    return None


def _eval_coef(bn, dev):
    """scale | shift of an eval-mode norm, kept on the module until one of its four tensors changes (version counters / storage): the
    frozen teacher's 26 norms and the student's no-grad frame re-derived them with a 5 us launch per call"""
    key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version, bn.weight.data_ptr(),
           bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.eps), str(dev))
    hit = bn.__dict__.get("_dbev_eval_coef")
    if hit is None or hit[0] != key:
        # (a stale entry's tensor is written again, not replaced: a captured hipGraph may read through its address -- graphed.py)
        reuse = hit is not None and hit[1].numel() == 2 * bn.num_features and hit[1].device == torch.device(dev)
        coef = hit[1] if reuse else torch.empty((2 * bn.num_features,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.call("dbev_bn_infer_coef", L.ptr(bn.weight), L.ptr(bn.bias), L.ptr(bn.running_mean), L.ptr(bn.running_var),
                   float(bn.eps), bn.num_features, L.ptr(coef), L.stream_ptr(dev))
        hit = (key, coef, L.fingerprint(bn.weight, bn.bias, bn.running_mean, bn.running_var))
        bn.__dict__["_dbev_eval_coef"] = hit
    else:
        L.check_fingerprint(hit[2], "eval-mode norm coefficients", bn.weight, bn.bias, bn.running_mean, bn.running_var)
    L.note_derived("eval_coef", bn, (dev,), (hit[1],))
    return hit[1]


def invalidate_eval_coef(root):
    """forget the kept eval-mode coefficients of every norm under `root` (and the packed / folded Winograd filters of its convolutions).  The cache follows version counters, which writes through
    `.data` (EMA hooks, mmcv-style `.data` loads, collectives on `t.data`) do NOT move: code that updates norm tensors that way calls
    this afterwards (GradReducer's construction-time broadcast bumps the versions itself)."""
    for m in root.modules():
        m.__dict__.pop("_dbev_eval_coef", None)
        m.__dict__.pop("_dbev_wino_folded", None)          # filters packed with a folded norm (wino.conv_norm_relu_eval)
        w = getattr(m, "weight", None)
        if w is not None and hasattr(w, "_dbev_wino_pair"):  # packed Winograd filters kept on the weight (wino.packed_pair)
            del w._dbev_wino_pair
        if w is not None and hasattr(w, "_dbev_bf6_packs"):  # bf16 planes of a 1x1 filter (gemm_bf6.packed)
            del w._dbev_bf6_packs


def _infer(x, residual, bn, relu):
    dev = x.device
    N, C, H, W = x.shape
    y = torch.empty_like(x)
    coef = _eval_coef(bn, dev)
    with torch.cuda.device(dev):
        L.call("dbev_bn_act_apply", L.ptr(x), L.ptr(residual), L.ptr(coef), int(relu), L.ptr(y), N * H * W, C, L.stream_ptr(dev))
    return y


def bn_act(x, bn, residual=None, relu=True, pre=None, fork=False):
    """relu(bn(x) + residual) with the module `bn`'s parameters, statistics and mode.  `pre`: partial statistics rows of x taken by
    its producer (conv1x1_stats) -- only handed over when the training-mode kernel path applies (conv1x1_bn_ready).  `fork`: the
    result is a residual block's output; under autograd it then carries a second handle (`forked(y)`) for the next block's identity
    branch (see _BNActTrain.forward)."""
    if eligible(x, bn, residual):
        if bn.training or bn.running_mean is None:
            # num_batches_tracked += 1 happens inside the finalize kernel (no extra launch)
            fk = bool(fork and _state.get("fork", True) and torch.is_grad_enabled())
            return _fork_out(_BNActTrain.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                               bn.num_batches_tracked, bn.momentum, bn.eps, relu, pre, fk), fk)
        return _infer(x, residual, bn, relu)
    if x.is_cuda and _state["enabled"] and x.numel() > 0:
        L.note_fallback("bn_act", _why_not(x, bn, residual))
    out = nn.BatchNorm2d.forward(bn, x) if isinstance(bn, BatchNormAct2d) else bn(x)
    if residual is not None:
        out = out + residual
    return F.relu(out) if relu else out


class BatchNormAct2d(nn.BatchNorm2d):
    """nn.BatchNorm2d that also applies the ReLU which followed it in the reference's module list
    (parameters, buffers and state-dict keys are those of the BatchNorm2d it replaces)."""

    def forward(self, x):
        return bn_act(x, self, None, True)

    def extra_repr(self):
        return super().extra_repr() + ", fused_act=ReLU"


_BN_TYPES = (nn.BatchNorm2d, BatchNormAct2d)


def _is_plain_relu(m):
    return type(m) is nn.ReLU


def fuse_bn_relu_modules(model):
    """Rewire every `BatchNorm2d -> ReLU` module pair of nn.Sequential stacks and mmcv-style ConvModules.
    Returns the number of fused pairs.  Idempotent."""
    from .registry import ConvModule
    n = 0
    for mod in model.modules():
        if isinstance(mod, nn.Sequential):
            kids = list(mod._modules.items())
            for (ka, a), (kb, b) in zip(kids, kids[1:]):
                if type(a) is nn.BatchNorm2d and _is_plain_relu(b):
                    a.__class__ = BatchNormAct2d
                    mod._modules[kb] = nn.Identity()
                    n += 1
        elif isinstance(mod, ConvModule) and mod.with_norm and mod.with_activation:
            if type(mod.norm) is nn.BatchNorm2d and _is_plain_relu(mod.activate):
                mod.norm.__class__ = BatchNormAct2d
                mod.activate = nn.Identity()
                n += 1
    return n
