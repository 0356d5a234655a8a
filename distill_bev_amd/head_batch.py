"""The branch stacks of CenterHead evaluated together.

Every SeparateHead of the reference (mmdet3d/models/dense_heads/centerpoint_head.py:17-130) is a dict of small stacks
`ConvModule(C -> Ch, 3x3, BN, ReLU) -> Conv2d(Ch -> 1..3, 3x3)`, and CenterHead (:233-363) runs 6 tasks x 6 branches = 36 of
them on the SAME shared feature map: 36 convolutions C -> Ch, 36 BatchNorms over 33 MB maps, 36 weight-gradient and 36
data-gradient convolutions whose results autograd then sums with 35 full-map additions.  Per output channel that is the
arithmetic of ONE convolution C -> 36*Ch and ONE BatchNorm over 36*Ch channels, which is how it runs here:

* the first-layer weights of a group of branches are concatenated along the output-channel axis (a 5 MB copy) and applied in one
  convolution; its data gradient IS the sum over the group's branches;
* one fused BatchNorm + ReLU pass (bn_act kernels) normalises the whole group -- batch statistics are per channel, so every
  branch sees exactly its own; running statistics are written back to the 36 modules with two multi-tensor copies;
* the final convolutions (skinny kernels) read their 64 channels in place from the wide map and write their slice of the one shared
  gradient tensor in place -- all branches of a group in one launch each way (`dbev_skinny_conv3x3_multi_*`, grid.y = branch).

Groups: the fused norm kernels take C/4 = 2^k <= 256 or a multiple of 256 channels columns, so the 36 x 64 channels of the recipe
split into a group of 32 branches (2048 channels) and one of 4 (256).  Parameters, buffers and state-dict keys are untouched;
`plan_branches(head)` only records which modules belong together, and CenterHead falls back to the per-branch module calls
whenever the plan does not apply (eval mode, no_grad, NCHW inputs, `only=` pruning of the frozen teacher, fused norms disabled).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib as L
from . import bn_act as BA
from . import wino as W3
from .skinny_conv import SkinnyConv2d


class _BranchFinalConvs(Function):
    """y_i = conv3x3(A[:, i*Ch:(i+1)*Ch], w_i) + b_i for the branches of one group: ONE launch forward, data gradient and weight
    gradient each (dbev_skinny_conv3x3_multi_*, grid.y = branch), reading / writing the channel slices of the wide tensors in place."""

    @staticmethod
    def forward(ctx, A, Ch, *wb):
        dev = A.device
        N, Ct, H, W = A.shape
        n = len(wb) // 2
        assert n * Ch == Ct
        couts = [int(w.shape[0]) for w in wb[0::2]]
        pad = torch.zeros((3, 9 * Ch), dtype=torch.float32, device=dev)     # zero rows: weight padding and the bias of a bias-free branch
        wrows, brows = [], []
        for i in range(n):
            w, b = wb[2 * i], wb[2 * i + 1]
            wrows.append(w.permute(0, 2, 3, 1).reshape(couts[i], 9 * Ch))
            brows.append(b if b is not None else pad[:couts[i], 0])
            if couts[i] < 3:
                wrows.append(pad[:3 - couts[i]]); brows.append(pad[:3 - couts[i], 0])
        wpk = torch.cat(wrows)                          # [n * 3, 9 * Ch]: every branch padded to 3 output rows
        bpk = torch.cat(brows)
        ys = [torch.empty((N, co, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last) for co in couts]
        with torch.cuda.device(dev):
            L.call("dbev_skinny_conv3x3_multi_forward", L.ptr(A), Ct, L.ptr(wpk), L.ptr(bpk), L.host_ptrs(ys), L.host_ints(couts), n,
                   N, Ch, H, W, L.stream_ptr(dev), alg_bytes=4 * N * H * W * (Ct + sum(couts)))
        ctx.save_for_backward(A, wpk)
        ctx.cfg = (Ch, couts, [b is not None for b in wb[1::2]])
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        A, wpk = ctx.saved_tensors
        Ch, couts, has_bias = ctx.cfg
        dev = A.device
        N, Ct, H, W = A.shape
        n = len(couts)
        gys = [(g if g is not None else torch.zeros((N, co, H, W), dtype=torch.float32, device=dev)).contiguous(memory_format=torch.channels_last)
               for g, co in zip(gys, couts)]
        gA = torch.empty_like(A)                      # every channel slice is fully written by its branch
        gw = torch.empty((n, 3, 9, Ch), dtype=torch.float32, device=dev)
        gb = torch.empty((n, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nbytes = int(L.call("dbev_skinny_conv3x3_multi_workspace_bytes", Ch, n))
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            L.call("dbev_skinny_conv3x3_multi_backward", L.host_ptrs(gys), L.ptr(A), Ct, L.ptr(wpk), L.ptr(gA), Ct, L.ptr(gw), L.ptr(gb),
                   L.host_ints(couts), n, N, Ch, H, W, L.ptr(ws), ws.numel(), L.stream_ptr(dev),
                   alg_bytes=4 * N * H * W * (2 * Ct + 2 * sum(couts)))
        grads = []
        for i, co in enumerate(couts):
            grads += [gw[i, :co].view(co, 3, 3, Ch).permute(0, 3, 1, 2), gb[i, :co] if has_bias[i] else None]
        return (gA, None, *grads)


def _branch_ok(seq):
    from .registry import ConvModule
    if not (isinstance(seq, nn.Sequential) and len(seq) == 2):
        return False
    cm, fin = seq[0], seq[1]
    if not (isinstance(cm, ConvModule) and cm.with_norm and isinstance(cm.norm, BA.BatchNormAct2d) and isinstance(cm.activate, nn.Identity)):
        return False
    c = cm.conv
    if not (type(c) in (nn.Conv2d, W3.WinoConv2d) and c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.dilation == (1, 1)
            and c.groups == 1 and c.bias is None and c.padding_mode == "zeros"):
        return False
    bn = cm.norm
    if not (bn.affine and bn.track_running_stats and bn.momentum is not None):
        return False
    return isinstance(fin, SkinnyConv2d) and fin.in_channels == c.out_channels


def plan_branches(head):
    """Record on `head` (a CenterHead whose norms / final convolutions already run on the fused kernels) which branch stacks can be
    evaluated together; returns the number of branches covered (0: the head keeps the per-branch calls)."""
    head._branch_plan = None
    branches = [(t, name, getattr(task, name)) for t, task in enumerate(head.task_heads) for name in task.heads]
    if len(branches) < 2 or not all(_branch_ok(s) for _, _, s in branches):
        return 0
    c0 = branches[0][2][0].conv
    bn0 = branches[0][2][0].norm
    Ch, Cin = c0.out_channels, c0.in_channels
    if Ch % 4 or (Ch // 4) not in (8, 16, 32, 64):
        return 0
    for _, _, s in branches:
        c, bn = s[0].conv, s[0].norm
        if (c.out_channels, c.in_channels) != (Ch, Cin) or bn.eps != bn0.eps or bn.momentum != bn0.momentum:
            return 0
    groups, i = [], 0
    while i < len(branches):
        n = len(branches) - i
        while n > 1 and not BA._channels_ok(n * Ch):
            n -= 1
        groups.append(branches[i:i + n])
        i += n
    head._branch_plan = dict(Ch=Ch, groups=groups)
    return len(branches)


def applies(head, x):
    plan = getattr(head, "_branch_plan", None)
    ok = _applies(head, x, plan)
    if not ok and plan is not None and head.training and torch.is_grad_enabled() and BA._state["enabled"] and x.is_cuda:
        L.note_fallback("head_batch", "not channels-last" if not BA._nhwc(x) else "branch norm in eval mode / shape")
    return ok


def _applies(head, x, plan):
    return (plan is not None and head.training and torch.is_grad_enabled() and BA._state["enabled"] and x.is_cuda
            and x.dtype == torch.float32 and BA._nhwc(x) and x.numel() > 0 and x.shape[0] * x.shape[2] * x.shape[3] > 1
            # a branch norm put into eval mode on its own (frozen statistics) needs its own path: per-branch calls
            and all(s[0].norm.training for group in plan["groups"] for _, _, s in group))


def forward(head, x):
    """-> list over tasks of {branch name: tensor}, as [task(x) for task in head.task_heads]"""
    plan = head._branch_plan
    Ch = plan["Ch"]
    dev = x.device
    outs = [dict() for _ in head.task_heads]
    for group in plan["groups"]:
        mods = [s[0] for _, _, s in group]
        bns = [m.norm for m in mods]
        w = torch.cat([m.conv.weight for m in mods], 0)
        rows = None
        if W3.eligible(x, w) and W3.worthwhile(x, w.shape[0]):      # Winograd kernels: the group norm's statistics come out of the epilogue
            a, rows = W3.conv3x3_stats(x, w, None)
        else:
            a = F.conv2d(x, w, None, 1, 1)
        gamma = torch.cat([bn.weight for bn in bns])
        beta = torch.cat([bn.bias for bn in bns])
        rm = torch.cat([bn.running_mean for bn in bns])
        rv = torch.cat([bn.running_var for bn in bns])
        nbt = torch.zeros((), dtype=torch.long, device=dev)
        y = BA._BNActTrain.apply(a, None, gamma, beta, rm, rv, nbt, bns[0].momentum, bns[0].eps, True, rows)
        with torch.no_grad():
            torch._foreach_copy_([bn.running_mean for bn in bns], list(rm.split(Ch)))
            torch._foreach_copy_([bn.running_var for bn in bns], list(rv.split(Ch)))
            torch._foreach_add_([bn.num_batches_tracked for bn in bns], 1)
        wb = []
        for _, _, s in group:
            wb += [s[1].weight, s[1].bias]
        ys = _BranchFinalConvs.apply(y, Ch, *wb)
        for (t, name, _), yi in zip(group, ys):
            outs[t][name] = yi
    # the reference's dict order (common heads, then heatmap)
    return [{name: outs[t][name] for name in task.heads} for t, task in enumerate(head.task_heads)]
