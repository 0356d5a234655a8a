"""CenterHead -- registry mirror of ``mmdet3d/models/dense_heads/centerpoint_head.py``
(SeparateHead :17-130, CenterHead :246-686) with the mmdet==2.24.0 losses it is configured with
(GaussianFocalLoss, L1Loss, MSELoss; un-vendored, standard definitions) and the target
assignment of ``get_targets_single`` :447-611 / ``core/utils/gaussian.py`` :6-88.

The dense head stays PyTorch (MIOpen convs).  Target assignment (in the reference a python loop over the GT
boxes issuing hundreds of tiny device ops per sample, plus ``.item()`` syncs) is ONE library call for all
tasks and samples (``dbev_centerhead_targets``, csrc/center_targets.hip) after two uploads (boxes, labels).
"""
import copy

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from .registry import BBOX_CODERS, MODELS, ConvModule, build_conv_layer, build_head, build_loss


class LiDARBoxes:
    """The two members of ``LiDARInstance3DBoxes`` the hot path touches
    (``core/bbox/structures/lidar_box3d.py:41-47``): ``tensor`` f32[M, 9] =
    (x, y, z_bottom, w, l, h, yaw, vx, vy) on the host, and ``gravity_center``."""

    def __init__(self, tensor):
        t = torch.as_tensor(tensor, dtype=torch.float32)
        self.tensor = t.reshape(-1, t.shape[-1] if t.numel() else 9)

    @property
    def gravity_center(self):
        gc = self.tensor[:, :3].clone()
        gc[:, 2] = gc[:, 2] + self.tensor[:, 5] * 0.5
        return gc

    def __len__(self):
        return self.tensor.shape[0]


def clip_sigmoid(x, eps=1e-4):
    """models/utils/clip_sigmoid.py:5-19 (in-place sigmoid, as the reference)."""
    return torch.clamp(x.sigmoid_(), min=eps, max=1 - eps)


# ---- losses (mmdet 2.24 semantics) -------------------------------------------------------
def _weight_reduce(loss, weight, reduction, avg_factor):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean() if reduction == "mean" else loss.sum() if reduction == "sum" else loss
    if reduction == "mean":
        return loss.sum() / avg_factor
    if reduction == "none":
        return loss
    raise ValueError("avg_factor can not be used with reduction='sum'")


@MODELS.register_module()
class GaussianFocalLoss(nn.Module):
    def __init__(self, alpha=2.0, gamma=4.0, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.alpha, self.gamma, self.reduction, self.loss_weight = alpha, gamma, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        eps = 1e-12
        pos_w = target.eq(1)
        neg_w = (1 - target).pow(self.gamma)
        pos = -(pred + eps).log() * (1 - pred).pow(self.alpha) * pos_w
        neg = -(1 - pred + eps).log() * pred.pow(self.alpha) * neg_w
        return self.loss_weight * _weight_reduce(pos + neg, weight, reduction_override or self.reduction, avg_factor)


@MODELS.register_module()
class L1Loss(nn.Module):
    def __init__(self, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        if target.numel() == 0:
            return pred.sum() * 0
        return self.loss_weight * _weight_reduce(torch.abs(pred - target), weight,
                                                 reduction_override or self.reduction, avg_factor)


@MODELS.register_module()
class MSELoss(nn.Module):
    def __init__(self, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        return self.loss_weight * _weight_reduce((pred - target) ** 2, weight,
                                                 reduction_override or self.reduction, avg_factor)


@MODELS.register_module()
class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        d = torch.abs(pred - target)
        loss = torch.where(d < self.beta, 0.5 * d * d / self.beta, d - 0.5 * self.beta)
        return self.loss_weight * _weight_reduce(loss, weight, reduction_override or self.reduction, avg_factor)


@BBOX_CODERS.register_module()
class CenterPointBBoxCoder:
    """core/bbox/coders/centerpoint_bbox_coders.py -- only the config surface is needed for
    training (decode belongs to inference post-processing, out of scope)."""

    def __init__(self, pc_range, out_size_factor, voxel_size, post_center_range=None, max_num=100,
                 score_threshold=None, code_size=9):
        self.pc_range, self.out_size_factor, self.voxel_size = pc_range, out_size_factor, voxel_size
        self.post_center_range, self.max_num, self.score_threshold, self.code_size = \
            post_center_range, max_num, score_threshold, code_size


# ---- heads ----------------------------------------------------------------------------
@MODELS.register_module()
class SeparateHead(nn.Module):
    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, init_bias=-2.19,
                 conv_cfg=dict(type="Conv2d"), norm_cfg=dict(type="BN2d"), bias="auto", init_cfg=None,
                 act_cfg=dict(type="ReLU"), **kwargs):
        super().__init__()
        self.heads = heads
        self.init_bias = init_bias
        for head, (classes, num_conv) in heads.items():
            layers = []
            c_in = in_channels
            for _ in range(num_conv - 1):
                layers.append(ConvModule(c_in, head_conv, kernel_size=final_kernel, stride=1,
                                         padding=final_kernel // 2, bias=bias, conv_cfg=conv_cfg,
                                         norm_cfg=norm_cfg, act_cfg=act_cfg))
                c_in = head_conv
            layers.append(build_conv_layer(conv_cfg, head_conv, classes, kernel_size=final_kernel, stride=1,
                                           padding=final_kernel // 2, bias=True))
            self.add_module(head, nn.Sequential(*layers))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        if "heatmap" in heads:
            getattr(self, "heatmap")[-1].bias.data.fill_(init_bias)

    def forward(self, x, only=None):
        """``only``: names of the branches to evaluate (default all) -- the distillation reads nothing but the teacher's heat maps"""
        return {head: getattr(self, head)(x) for head in self.heads if only is None or head in only}



_HEAD_ORDER = ("reg", "height", "dim", "rot", "vel", "heatmap")


def _nhwc_flag(t):
    """1 if the tensor is physically NHWC (and not also NCHW-contiguous); raises if it is neither."""
    if t.is_contiguous():
        return 0
    if t.is_contiguous(memory_format=torch.channels_last):
        return 1
    raise ValueError("head output must be NCHW- or channels-last-contiguous")


class _CenterHeadLoss(torch.autograd.Function):
    """All tasks' heat-map focal losses and task-specific L1 terms on the HIP kernels (csrc/center_loss.hip).
    forward(targets..., *36 head tensors) -> (losses f32[T, 6], clipped-sigmoid heat map per task)."""

    @staticmethod
    def forward(ctx, hm, anno, ind, mask, ncls, code_w, lw_bbox, lw_cls, *heads):
        from . import _lib as L
        T = len(ncls)
        dev = heads[0].device
        heads = [h if (h.is_contiguous() or h.is_contiguous(memory_format=torch.channels_last)) else h.contiguous()
                 for h in heads]
        flags = [_nhwc_flag(h) for h in heads]
        B, _, H, W = heads[5].shape
        sig = [torch.empty_like(heads[t * 6 + 5]) for t in range(T)]
        losses = torch.empty((T, 6), dtype=torch.float32, device=dev)
        avg = torch.empty((2 * T,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nbytes = L.call("dbev_centerhead_loss_workspace_bytes", B, int(sum(ncls)), T, H, W)
            ws = torch.empty((int(nbytes),), dtype=torch.uint8, device=dev)
            L.call("dbev_centerhead_loss_forward", L.host_ptrs(heads), L.host_ints(flags), L.host_ptrs(sig),
                   L.host_ints(ncls), T, B, H, W, anno.shape[2], L.ptr(hm), L.ptr(anno), L.ptr(ind), L.ptr(mask),
                   L.host_floats(code_w), float(lw_bbox), float(lw_cls), L.ptr(losses), L.ptr(avg), L.ptr(ws),
                   ws.numel(), L.stream_ptr(dev))
        ctx.save_for_backward(hm, anno, ind, mask, avg, *heads)
        ctx.cfg = (tuple(ncls), tuple(code_w), float(lw_bbox), float(lw_cls), tuple(flags), (B, H, W))
        ctx.mark_non_differentiable(*sig)
        ctx.set_materialize_grads(False)                  # no zero maps for the (never used) gradients of the sigmoid outputs
        return (losses, *sig)

    @staticmethod
    def backward(ctx, g_losses, *_):
        from . import _lib as L
        hm, anno, ind, mask, avg, *heads = ctx.saved_tensors
        ncls, code_w, lw_bbox, lw_cls, flags, (B, H, W) = ctx.cfg
        T = len(ncls)
        if g_losses is None:
            return (None,) * (8 + len(heads))
        dev = g_losses.device
        # one zero-filled slab for the 5 regression-head gradients of every task (only object pixels are written),
        # heat-map gradients are fully written by the kernel
        sizes = [h.numel() for h in heads]
        reg_total = sum(n for i, n in enumerate(sizes) if i % 6 != 5)
        slab = torch.zeros((reg_total,), dtype=torch.float32, device=dev)
        grads, o = [], 0
        for i, h in enumerate(heads):
            if i % 6 == 5:
                grads.append(torch.empty_like(h))
            else:
                g = slab[o:o + sizes[i]]
                o += sizes[i]
                grads.append(g.view(h.shape[0], h.shape[2], h.shape[3], h.shape[1]).permute(0, 3, 1, 2) if flags[i]
                             else g.view(h.shape))
        with torch.cuda.device(dev):
            L.call("dbev_centerhead_loss_backward", L.host_ptrs(heads), L.host_ints(flags), L.host_ptrs(grads),
                   L.host_ints(ncls), T, B, H, W, anno.shape[2], L.ptr(hm), L.ptr(anno), L.ptr(ind), L.ptr(mask),
                   L.host_floats(code_w), lw_bbox, lw_cls, L.ptr(avg), L.ptr(g_losses.contiguous().float()),
                   L.stream_ptr(dev))
        return (None,) * 8 + tuple(grads)


@MODELS.register_module()
class CenterHead(nn.Module):
    def __init__(self, in_channels=[128], tasks=None, train_cfg=None, test_cfg=None, bbox_coder=None,
                 common_heads=dict(), loss_cls=dict(type="GaussianFocalLoss", reduction="mean"),
                 loss_bbox=dict(type="L1Loss", reduction="none", loss_weight=0.25),
                 separate_head=dict(type="SeparateHead", init_bias=-2.19, final_kernel=3),
                 share_conv_channel=64, num_heatmap_convs=2, conv_cfg=dict(type="Conv2d"),
                 norm_cfg=dict(type="BN2d"), bias="auto", norm_bbox=True, init_cfg=None, task_specific=True,
                 loss_prefix="", act_cfg=dict(type="ReLU")):
        super().__init__()
        num_classes = [len(t["class_names"]) for t in tasks]
        self.class_names = [t["class_names"] for t in tasks]
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.in_channels, self.num_classes, self.norm_bbox = in_channels, num_classes, norm_bbox
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.bbox_coder = BBOX_CODERS.build(bbox_coder) if bbox_coder is not None else None
        self.shared_conv = ConvModule(in_channels, share_conv_channel, kernel_size=3, padding=1,
                                      conv_cfg=conv_cfg, norm_cfg=norm_cfg, bias=bias, act_cfg=act_cfg)
        self.task_heads = nn.ModuleList()
        for num_cls in num_classes:
            heads = copy.deepcopy(dict(common_heads))
            heads.update(dict(heatmap=(num_cls, num_heatmap_convs)))
            sh = dict(separate_head)
            sh.update(in_channels=share_conv_channel, heads=heads, num_cls=num_cls)
            self.task_heads.append(build_head(sh))
        self.task_specific, self.loss_prefix = task_specific, loss_prefix

    def forward_single(self, x, only=None):
        x = self.shared_conv(x)
        if only is None:
            from . import head_batch
            if head_batch.applies(self, x):          # the 36 branch stacks as a few wide convolution / norm calls
                return head_batch.forward(self, x)
        return [task(x) if only is None else task(x, only) for task in self.task_heads]

    def forward(self, feats, only=None):
        """-> tuple over tasks of [dict] (multi_apply transposition, centerpoint_head.py:352-363)."""
        per_level = [self.forward_single(f, only) for f in feats]
        return tuple([lvl[t] for lvl in per_level] for t in range(len(self.task_heads)))

    # ---- targets -------------------------------------------------------------------------
    def get_targets_device(self, gt_bboxes_3d, gt_labels_3d, device):
        """:366-413 on the GPU: two uploads (boxes, labels), one library call (dbev_centerhead_targets) for all
        tasks and samples; returns the same per-task lists as the host path (views of the packed outputs)."""
        from . import _lib as L
        cfg = self.train_cfg
        B, T = len(gt_bboxes_3d), len(self.task_heads)
        b9, labs, starts = [], [], [0]
        for boxes, labels in zip(gt_bboxes_3d, gt_labels_3d):
            t9 = torch.cat((boxes.gravity_center, boxes.tensor[:, 3:]), dim=1).float()
            b9.append(t9.cpu()); labs.append(torch.as_tensor(labels).to(torch.int32).cpu())
            starts.append(starts[-1] + t9.shape[0])
        boxes_d = L.h2d(torch.cat(b9).contiguous(), device) if starts[-1] else torch.zeros((1, 9), device=device)
        labels_d = L.h2d(torch.cat(labs).contiguous(), device) if starts[-1] else torch.zeros((1,), dtype=torch.int32, device=device)
        max_objs = cfg["max_objs"] * cfg["dense_reg"]
        osf = cfg["out_size_factor"]
        W, H = int(cfg["grid_size"][0]) // osf, int(cfg["grid_size"][1]) // osf
        ncls = [len(n) for n in self.class_names]
        hm = torch.empty((B, sum(ncls), H, W), dtype=torch.float32, device=device)
        ab = torch.empty((T, B, max_objs, 10), dtype=torch.float32, device=device)
        ind = torch.empty((T, B, max_objs), dtype=torch.int64, device=device)
        mk = torch.empty((T, B, max_objs), dtype=torch.uint8, device=device)
        ws = torch.empty((16 * max(starts[-1], 1),), dtype=torch.uint8, device=device)
        pc, vs = cfg["point_cloud_range"], cfg["voxel_size"]
        with torch.cuda.device(device):
            L.call("dbev_centerhead_targets", L.ptr(boxes_d), L.ptr(labels_d), L.host_ints(starts), B, L.host_ints(ncls), T,
                   H, W, max_objs, int(cfg["min_radius"]), float(cfg["gaussian_overlap"]), float(pc[0]), float(pc[1]),
                   float(vs[0]), float(vs[1]), int(osf), 1 if self.norm_bbox else 0, L.ptr(hm), L.ptr(ab), L.ptr(ind),
                   L.ptr(mk), L.ptr(ws), ws.numel(), L.stream_ptr(device))
        edges = np.cumsum([0] + ncls)
        self._packed_targets = (hm, ab, ind, mk)       # the fused loss consumes the packed tensors directly
        return ([hm[:, edges[t]:edges[t + 1]] for t in range(T)], [ab[t] for t in range(T)],
                [ind[t] for t in range(T)], [mk[t] for t in range(T)])

    def get_targets(self, gt_bboxes_3d, gt_labels_3d, device):
        """:366-413 -> per task lists (heatmaps, anno_boxes, inds, masks) on `device`.  GPU only: the host restatement
        the kernels are tested against lives in oracle/center_targets.py (test infrastructure)."""
        if torch.device(device).type != "cuda":
            from ._lib import DbevHipError
            raise DbevHipError("CenterHead.get_targets runs on the HIP kernels (dbev_centerhead_targets) and needs a "
                               "GPU device; there is no CPU fallback in the product path")
        return self.get_targets_device(gt_bboxes_3d, gt_labels_3d, device)

    fused_loss = True      # class-level switch (tests compare against the op-by-op sequence below)

    def _fused_loss_ok(self, preds_dicts, heatmaps):
        """the HIP loss covers the recipe's configuration: task-specific L1 groups, GaussianFocalLoss(2, 4, mean),
        L1Loss(mean), packed device targets, fp32 head outputs on the GPU"""
        p0 = preds_dicts[0][0]
        pk = getattr(self, "_packed_targets", None)
        packed = pk is not None and getattr(heatmaps[0], "_base", None) is pk[0]
        return (self.fused_loss and self.task_specific and p0["heatmap"].is_cuda and p0["heatmap"].dtype == torch.float32
                and packed and type(self.loss_cls) is GaussianFocalLoss and self.loss_cls.alpha == 2.0
                and self.loss_cls.gamma == 4.0 and self.loss_cls.reduction == "mean" and type(self.loss_bbox) is L1Loss
                and self.loss_bbox.reduction == "mean" and self.train_cfg.get("code_weights", None) is not None
                and all(set(_HEAD_ORDER) <= set(p[0]) for p in preds_dicts))

    def _fused_loss(self, preds_dicts, heatmaps, anno_boxes, inds, masks):
        """:615-686 for all tasks through dbev_centerhead_loss_* (side effect kept: p['heatmap'] becomes the clipped
        sigmoid, which add_fp_as_fg reads later)."""
        T = len(preds_dicts)
        hm, anno, ind, mask = self._packed_targets
        ncls = [len(n) for n in self.class_names]
        heads = [preds_dicts[t][0][k] for t in range(T) for k in _HEAD_ORDER]
        out = _CenterHeadLoss.apply(hm, anno, ind, mask, ncls, [float(v) for v in self.train_cfg["code_weights"]],
                                    self.loss_bbox.loss_weight, self.loss_cls.loss_weight, *heads)
        sig = out[1:]
        vals = out[0].reshape(-1).unbind(0)          # ONE backward node (a stack) for the 36 scalars
        loss_dict = dict()
        names = ["xy", "z", "whl", "yaw", "vel"]
        for t in range(T):
            preds_dicts[t][0]["heatmap"] = sig[t]
            for r, nm in enumerate(names):
                loss_dict[f"{self.loss_prefix}task{t}.loss_{nm}"] = vals[t * 6 + r]
            loss_dict[f"{self.loss_prefix}task{t}.loss_heatmap"] = vals[t * 6 + 5]
        return loss_dict

    @staticmethod
    def _gather_feat(feat, ind):
        dim = feat.size(2)
        return feat.gather(1, ind.unsqueeze(2).expand(ind.size(0), ind.size(1), dim))

    def loss(self, gt_bboxes_3d, gt_labels_3d, preds_dicts, get_targets=False, **kwargs):
        """:615-686 (clip_sigmoid is applied IN PLACE to the predicted heatmaps, as the reference)."""
        device = preds_dicts[0][0]["heatmap"].device
        heatmaps, anno_boxes, inds, masks = self.get_targets(gt_bboxes_3d, gt_labels_3d, device)
        if self._fused_loss_ok(preds_dicts, heatmaps):
            loss_dict = self._fused_loss(preds_dicts, heatmaps, anno_boxes, inds, masks)
            return (loss_dict, heatmaps, anno_boxes, inds, masks) if get_targets else loss_dict
        loss_dict = dict()
        code_weights = self.train_cfg.get("code_weights", None)
        for task_id, preds_dict in enumerate(preds_dicts):
            p = preds_dict[0]
            p["heatmap"] = clip_sigmoid(p["heatmap"])
            num_pos = heatmaps[task_id].eq(1).float().sum()
            loss_heatmap = self.loss_cls(p["heatmap"], heatmaps[task_id], avg_factor=torch.clamp(num_pos, min=1))
            target_box = anno_boxes[task_id]
            p["anno_box"] = torch.cat((p["reg"], p["height"], p["dim"], p["rot"], p["vel"]), dim=1)
            ind = inds[task_id]
            num = masks[task_id].float().sum()
            pred = p["anno_box"].permute(0, 2, 3, 1).contiguous()
            pred = self._gather_feat(pred.view(pred.size(0), -1, pred.size(3)), ind)
            mask = masks[task_id].unsqueeze(2).expand_as(target_box).float()
            mask = mask * (~torch.isnan(target_box)).float()
            bbox_weights = mask * L.h2d_like(mask, code_weights)
            if self.task_specific:
                names, clip = ["xy", "z", "whl", "yaw", "vel"], [0, 2, 3, 6, 8, 10]
                for r, nm in enumerate(names):
                    sl = slice(clip[r], clip[r + 1])
                    loss_dict[f"{self.loss_prefix}task{task_id}.loss_{nm}"] = self.loss_bbox(
                        pred[..., sl], target_box[..., sl], bbox_weights[..., sl], avg_factor=(num + 1e-4))
            else:
                loss_dict[f"task{task_id}.loss_bbox"] = self.loss_bbox(pred, target_box, bbox_weights,
                                                                       avg_factor=(num + 1e-4))
            loss_dict[f"{self.loss_prefix}task{task_id}.loss_heatmap"] = loss_heatmap
        if get_targets:
            return loss_dict, heatmaps, anno_boxes, inds, masks
        return loss_dict
