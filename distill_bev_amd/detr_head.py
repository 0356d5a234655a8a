"""Set-prediction heads of the BEVFormer student and the LidarFormer / MVPFormer teachers, and what their configs name
around them: ``BEVFormerHead`` (``mmdet3d/models/dense_heads/bevformer_head.py:21-520``), ``DGCNN3DHead``
(``dgcnn3d_head.py:17-510``), ``NMSFreeCoder`` (``core/bbox/coders/nms_free_coder.py:9-121``), ``HungarianAssigner3D``
(``core/bbox/assigners/hungarian_assigner_3d.py:16-130``), ``BBox3DL1Cost`` (``core/bbox/match_costs/match_cost.py:5-27``),
``normalize_bbox / denormalize_bbox`` (``core/bbox/util.py:4-53``), plus the un-vendored mmdet 2.24 pieces the two heads
inherit or build: the ``DETRHead`` constructor surface, ``FocalLoss`` (sigmoid), ``FocalLossCost``, ``IoUCost`` /
``GIoULoss`` (configured with weight 0 "for DETR compatibility", never evaluated), ``PseudoSampler``, ``reduce_mean``.

Both heads share one class body here (``_SetPredictionHead``): the reference's two files differ in the forward pass, the
argument order of ``_get_target_single`` and how the box loss is split -- those are the overridden methods.
"""
import copy

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from .registry import BBOX_CODERS, MODELS
from . import _lib as L
from .transformer import build_positional_encoding, build_transformer, inverse_sigmoid

_EPS32 = torch.finfo(torch.float32).eps


def normalize_bbox(bboxes, pc_range=None):
    """util.py:4-26: (cx, cy, cz, w, l, h, rot[, vx, vy]) -> (cx, cy, log w, log l, cz, log h, sin, cos[, vx, vy])"""
    parts = [bboxes[..., 0:1], bboxes[..., 1:2], bboxes[..., 3:4].log(), bboxes[..., 4:5].log(), bboxes[..., 2:3],
             bboxes[..., 5:6].log(), bboxes[..., 6:7].sin(), bboxes[..., 6:7].cos()]
    if bboxes.size(-1) > 7:
        parts += [bboxes[..., 7:8], bboxes[..., 8:9]]
    return torch.cat(parts, dim=-1)


def denormalize_bbox(nb, pc_range=None):
    """util.py:28-53"""
    rot = torch.atan2(nb[..., 6:7], nb[..., 7:8])
    parts = [nb[..., 0:1], nb[..., 1:2], nb[..., 4:5], nb[..., 2:3].exp(), nb[..., 3:4].exp(), nb[..., 5:6].exp(), rot]
    if nb.size(-1) > 8:
        parts += [nb[:, 8:9], nb[:, 9:10]]
    return torch.cat(parts, dim=-1)


def reduce_mean(tensor):
    """mmdet.core.utils.reduce_mean: the mean over the ranks of a process group (identity without one)"""
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return tensor


def _weight_reduce(loss, weight, reduction, avg_factor):
    """mmdet 2.24 weight_reduce_loss"""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean() if reduction == "mean" else loss.sum() if reduction == "sum" else loss
    if reduction == "mean":
        return loss.sum() / (avg_factor + _EPS32)
    if reduction == "none":
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


@MODELS.register_module()
class FocalLoss(nn.Module):
    """mmdet FocalLoss, sigmoid form: labels in [0, num_classes] with num_classes = background."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0, activated=False):
        super().__init__()
        assert use_sigmoid is True and not activated, "only the sigmoid focal loss on logits is configured"
        self.use_sigmoid, self.gamma, self.alpha, self.reduction, self.loss_weight = use_sigmoid, gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        num_classes = pred.size(1)
        t = F.one_hot(target, num_classes=num_classes + 1)[:, :num_classes].type_as(pred)
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        focal = (self.alpha * t + (1 - self.alpha) * (1 - t)) * pt.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(pred, t, reduction="none") * focal
        if weight is not None and weight.shape != loss.shape:
            weight = weight.view(-1, 1) if weight.size(0) == loss.size(0) else weight.view(loss.size(0), -1)
        return self.loss_weight * _weight_reduce(loss, weight, reduction_override or self.reduction, avg_factor)


@MODELS.register_module()
class GIoULoss(nn.Module):
    """Configured with loss_weight 0 'for DETR compatibility' (bevformer configs) and never called by these heads."""

    def __init__(self, eps=1e-6, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def forward(self, *args, **kwargs):
        raise NotImplementedError("the 2-D GIoU loss has no meaning for 3-D boxes; the reference never evaluates it")


@MODELS.register_module()
class FocalLossCost(object):
    """mmdet FocalLossCost: [num_query, num_gt] classification cost"""

    def __init__(self, weight=1.0, alpha=0.25, gamma=2, eps=1e-12, binary_input=False):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        p = cls_pred.sigmoid()
        neg = -(1 - p + self.eps).log() * (1 - self.alpha) * p.pow(self.gamma)
        pos = -(p + self.eps).log() * self.alpha * (1 - p).pow(self.gamma)
        return (pos[:, gt_labels] - neg[:, gt_labels]) * self.weight


@MODELS.register_module()
class BBox3DL1Cost(object):
    """match_cost.py:5-27"""

    def __init__(self, weight=1.0):
        self.weight = weight

    def __call__(self, bbox_pred, gt_bboxes):
        return torch.cdist(bbox_pred, gt_bboxes, p=1) * self.weight


@MODELS.register_module()
class IoUCost(object):
    """'Fake cost' of the reference's configs (weight 0.0): built, never called."""

    def __init__(self, iou_mode="giou", weight=1.0):
        self.iou_mode, self.weight = iou_mode, weight


class AssignResult(object):
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


@MODELS.register_module()
class HungarianAssigner3D(object):
    """hungarian_assigner_3d.py:16-130: one-to-one matching of queries to ground-truth boxes on focal + L1 cost (the cost
    matrix goes to the host for scipy's linear_sum_assignment, as in the reference)."""

    def __init__(self, cls_cost=dict(type="ClassificationCost", weight=1.0), reg_cost=dict(type="BBoxL1Cost", weight=1.0),
                 iou_cost=dict(type="IoUCost", weight=0.0), pc_range=None):
        self.cls_cost, self.reg_cost, self.iou_cost = MODELS.build(cls_cost), MODELS.build(reg_cost), MODELS.build(iou_cost)
        self.pc_range = pc_range

    def cost_matrix(self, bbox_pred, cls_pred, gt_bboxes, gt_labels):
        """[num_query, num_gt] matching cost on the device (:95-103)"""
        return self.cls_cost(cls_pred, gt_labels) + self.reg_cost(bbox_pred[:, :8], normalize_bbox(gt_bboxes, self.pc_range)[:, :8])

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, gt_bboxes_ignore=None, eps=1e-7):
        from scipy.optimize import linear_sum_assignment
        assert gt_bboxes_ignore is None
        num_gts, num_bboxes = gt_bboxes.size(0), bbox_pred.size(0)
        gt_inds = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        labels = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            if num_gts == 0:
                gt_inds[:] = 0
            return AssignResult(num_gts, gt_inds, None, labels=labels)
        rows, cols = linear_sum_assignment(self.cost_matrix(bbox_pred, cls_pred, gt_bboxes, gt_labels).detach().cpu())
        rows = L.h2d(torch.from_numpy(rows), bbox_pred.device)
        cols = L.h2d(torch.from_numpy(cols), bbox_pred.device)
        gt_inds[:] = 0
        gt_inds[rows] = cols + 1
        labels[rows] = gt_labels[cols]
        return AssignResult(num_gts, gt_inds, None, labels=labels)


class _Sampled(object):
    pass


@MODELS.register_module()
class PseudoSampler(object):
    """mmdet PseudoSampler: every assigned query is a positive, the rest negatives."""

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        r = _Sampled()
        r.pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        r.neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        r.pos_assigned_gt_inds = assign_result.gt_inds[r.pos_inds] - 1
        r.pos_gt_bboxes = gt_bboxes[r.pos_assigned_gt_inds.long(), :] if gt_bboxes.numel() else gt_bboxes.view(0, gt_bboxes.shape[-1])
        return r


@BBOX_CODERS.register_module()
class NMSFreeCoder(object):
    """nms_free_coder.py:9-121: top-``max_num`` (query, class) pairs of the last decoder layer, range-filtered"""

    def __init__(self, pc_range, voxel_size=None, post_center_range=None, max_num=100, score_threshold=None, num_classes=10):
        self.pc_range, self.voxel_size, self.post_center_range = pc_range, voxel_size, post_center_range
        self.max_num, self.score_threshold, self.num_classes = max_num, score_threshold, num_classes

    def encode(self):
        pass

    def decode_single(self, cls_scores, bbox_preds):
        scores, indexs = cls_scores.sigmoid().view(-1).topk(self.max_num)
        labels = indexs % self.num_classes
        bbox_index = torch.div(indexs, self.num_classes, rounding_mode="floor")
        boxes = denormalize_bbox(bbox_preds[bbox_index], self.pc_range)
        if self.post_center_range is None:
            raise NotImplementedError("Need to reorganize output as a batch, only support post_center_range is not None for now!")
        rng = torch.as_tensor(self.post_center_range, device=scores.device, dtype=boxes.dtype)
        mask = (boxes[..., :3] >= rng[:3]).all(1) & (boxes[..., :3] <= rng[3:]).all(1)
        if self.score_threshold is not None:
            keep, thr = scores > self.score_threshold, self.score_threshold
            while keep.sum() == 0:                 # :76-82: relax the threshold until something is kept
                thr *= 0.9
                if thr < 0.01:
                    keep = scores > -1
                    break
                keep = scores >= thr
            if self.score_threshold:
                mask = mask & keep
        return {"bboxes": boxes[mask], "scores": scores[mask], "labels": labels[mask]}

    def decode(self, preds_dicts):
        cls, box = preds_dicts["all_cls_scores"][-1], preds_dicts["all_bbox_preds"][-1]
        return [self.decode_single(cls[i], box[i]) for i in range(cls.size(0))]


def multi_apply(func, *args, **kwargs):
    results = list(map(lambda *a: func(*a, **kwargs), *args))
    return tuple(map(list, zip(*results)))


class _SetPredictionHead(nn.Module):
    """What mmdet's DETRHead constructor sets up for both heads + the shared target / loss / decoding code."""

    def __init__(self, num_classes, in_channels, num_query=100, num_reg_fcs=2, transformer=None, sync_cls_avg_factor=False,
                 positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                 loss_cls=dict(type="CrossEntropyLoss", bg_cls_weight=0.1, use_sigmoid=False, loss_weight=1.0, class_weight=1.0),
                 loss_bbox=dict(type="L1Loss", loss_weight=5.0), loss_iou=dict(type="GIoULoss", loss_weight=2.0),
                 train_cfg=dict(assigner=dict(type="HungarianAssigner", cls_cost=dict(type="ClassificationCost", weight=1.0),
                                              reg_cost=dict(type="BBoxL1Cost", weight=5.0),
                                              iou_cost=dict(type="IoUCost", iou_mode="giou", weight=2.0))),
                 test_cfg=dict(max_per_img=100), init_cfg=None, with_box_refine=False, as_two_stage=False, bbox_coder=None,
                 num_cls_fcs=2, bev_h=30, bev_w=30, code_size=10, **kwargs):
        super().__init__()
        assert not as_two_stage, "the two-stage variant is not configured by the reference's distillation configs"
        self.bev_h, self.bev_w, self.with_box_refine, self.as_two_stage = bev_h, bev_w, with_box_refine, as_two_stage
        self.code_size = code_size
        self.bbox_coder = BBOX_CODERS.build(bbox_coder)
        self.pc_range = self.bbox_coder.pc_range
        self.num_cls_fcs = num_cls_fcs - 1
        self.bg_cls_weight, self.sync_cls_avg_factor = 0, sync_cls_avg_factor
        if train_cfg:
            assigner = train_cfg["assigner"]
            assert loss_cls["loss_weight"] == assigner["cls_cost"]["weight"], "The classification weight for loss and matcher should be exactly the same."
            assert loss_bbox["loss_weight"] == assigner["reg_cost"]["weight"], "The regression L1 weight for loss and matcher should be exactly the same."
            assert loss_iou["loss_weight"] == assigner["iou_cost"]["weight"], "The regression iou weight for loss and matcher should be exactly the same."
            self.assigner = MODELS.build(assigner)
            self.sampler = PseudoSampler()
        self.num_query, self.num_classes, self.in_channels, self.num_reg_fcs = num_query, num_classes, in_channels, num_reg_fcs
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.loss_cls, self.loss_bbox, self.loss_iou = MODELS.build(loss_cls), MODELS.build(loss_bbox), MODELS.build(loss_iou)
        self.cls_out_channels = num_classes if self.loss_cls.use_sigmoid else num_classes + 1
        self.positional_encoding = build_positional_encoding(positional_encoding)
        self.transformer = build_transformer(transformer)
        self.embed_dims = self.transformer.embed_dims
        assert positional_encoding["num_feats"] * 2 == self.embed_dims
        self._init_layers()

    def _init_layers(self):
        """bevformer_head.py:74-110 == dgcnn3d_head.py:45-79"""
        cls_branch = []
        for _ in range(self.num_reg_fcs):
            cls_branch += [nn.Linear(self.embed_dims, self.embed_dims), nn.LayerNorm(self.embed_dims), nn.ReLU(inplace=True)]
        cls_branch.append(nn.Linear(self.embed_dims, self.cls_out_channels))
        reg_branch = []
        for _ in range(self.num_reg_fcs):
            reg_branch += [nn.Linear(self.embed_dims, self.embed_dims), nn.ReLU()]
        reg_branch.append(nn.Linear(self.embed_dims, self.code_size))
        fc_cls, reg_branch = nn.Sequential(*cls_branch), nn.Sequential(*reg_branch)
        num_pred = self.transformer.decoder.num_layers
        if self.with_box_refine:
            self.cls_branches = nn.ModuleList(copy.deepcopy(fc_cls) for _ in range(num_pred))
            self.reg_branches = nn.ModuleList(copy.deepcopy(reg_branch) for _ in range(num_pred))
        else:
            self.cls_branches = nn.ModuleList([fc_cls for _ in range(num_pred)])
            self.reg_branches = nn.ModuleList([reg_branch for _ in range(num_pred)])
        self.bev_embedding = nn.Embedding(self.bev_h * self.bev_w, self.embed_dims)
        self.query_embedding = nn.Embedding(self.num_query, self.embed_dims * 2)

    def init_weights(self):
        self.transformer.init_weights()
        if self.loss_cls.use_sigmoid:
            bias_init = float(-torch.log(torch.tensor((1 - 0.01) / 0.01)))           # mmcv bias_init_with_prob(0.01)
            for m in self.cls_branches:
                nn.init.constant_(m[-1].bias, bias_init)

    def _decode_layers(self, hs, init_reference, inter_references):
        """The per-decoder-layer class scores and boxes (bevformer_head.py:163-197 == dgcnn3d_head.py:120-150):
        (x, y) and z are offsets on the layer's reference point, mapped through a sigmoid into the point-cloud range."""
        classes, coords = [], []
        lo, hi = self.pc_range[:3], self.pc_range[3:]
        for lvl in range(hs.shape[0]):
            reference = inverse_sigmoid(init_reference if lvl == 0 else inter_references[lvl - 1])
            classes.append(self.cls_branches[lvl](hs[lvl]))
            tmp = self.reg_branches[lvl](hs[lvl])
            assert reference.shape[-1] == 3
            x = (tmp[..., 0:1] + reference[..., 0:1]).sigmoid() * (hi[0] - lo[0]) + lo[0]
            y = (tmp[..., 1:2] + reference[..., 1:2]).sigmoid() * (hi[1] - lo[1]) + lo[1]
            z = (tmp[..., 4:5] + reference[..., 2:3]).sigmoid() * (hi[2] - lo[2]) + lo[2]
            coords.append(torch.cat((x, y, tmp[..., 2:4], z, tmp[..., 5:]), -1))
        return torch.stack(classes), torch.stack(coords)

    # -- targets / loss ---------------------------------------------------------------------------------------------------
    def _get_target_single(self, cls_score, bbox_pred, gt_labels, gt_bboxes, gt_bboxes_ignore=None):
        """bevformer_head.py:209-262"""
        num_bboxes = bbox_pred.size(0)
        assign_result = self.assigner.assign(bbox_pred, cls_score, gt_bboxes, gt_labels, gt_bboxes_ignore)
        s = self.sampler.sample(assign_result, bbox_pred, gt_bboxes)
        labels = gt_bboxes.new_full((num_bboxes,), self.num_classes, dtype=torch.long)
        labels[s.pos_inds] = gt_labels[s.pos_assigned_gt_inds]
        label_weights = gt_bboxes.new_ones(num_bboxes)
        bbox_targets = torch.zeros_like(bbox_pred)[..., :self._target_dims(gt_bboxes)]
        bbox_weights = torch.zeros_like(bbox_pred)
        bbox_weights[s.pos_inds] = 1.0
        bbox_targets[s.pos_inds] = s.pos_gt_bboxes
        return labels, label_weights, bbox_targets, bbox_weights, s.pos_inds, s.neg_inds

    def get_targets(self, cls_scores_list, bbox_preds_list, gt_bboxes_list, gt_labels_list, gt_bboxes_ignore_list=None):
        assert gt_bboxes_ignore_list is None, "Only supports for gt_bboxes_ignore setting to None."
        ignore = [None for _ in cls_scores_list]
        labels, label_w, box_t, box_w, pos, neg = multi_apply(self._get_target_single, cls_scores_list, bbox_preds_list,
                                                              gt_labels_list, gt_bboxes_list, ignore)
        return labels, label_w, box_t, box_w, sum(i.numel() for i in pos), sum(i.numel() for i in neg)

    def loss_single(self, cls_scores, bbox_preds, gt_bboxes_list, gt_labels_list, gt_bboxes_ignore_list=None):
        """bevformer_head.py:316-386 / dgcnn3d_head.py:214-262: the loss of one decoder layer over the batch"""
        num_imgs = cls_scores.size(0)
        labels, label_w, box_t, box_w, num_pos, num_neg = self.get_targets(
            [cls_scores[i] for i in range(num_imgs)], [bbox_preds[i] for i in range(num_imgs)], gt_bboxes_list, gt_labels_list,
            gt_bboxes_ignore_list)
        labels, label_w, box_t, box_w = torch.cat(labels, 0), torch.cat(label_w, 0), torch.cat(box_t, 0), torch.cat(box_w, 0)
        cls_scores = cls_scores.reshape(-1, self.cls_out_channels)
        cls_avg_factor = num_pos * 1.0 + num_neg * self.bg_cls_weight
        if self.sync_cls_avg_factor:
            cls_avg_factor = reduce_mean(cls_scores.new_tensor([cls_avg_factor]))
        cls_avg_factor = max(cls_avg_factor, 1)
        loss_cls = self.loss_cls(cls_scores, labels, label_w, avg_factor=cls_avg_factor)
        num_pos = torch.clamp(reduce_mean(loss_cls.new_tensor([num_pos])), min=1).item()
        bbox_preds = bbox_preds.reshape(-1, bbox_preds.size(-1))
        targets = normalize_bbox(box_t, self.pc_range)
        ok = torch.isfinite(targets).all(dim=-1)
        return self._box_loss(loss_cls, bbox_preds[ok], targets[ok], box_w[ok], num_pos)

    def loss(self, gt_bboxes_list, gt_labels_list, preds_dicts, gt_bboxes_ignore=None, img_metas=None):
        """bevformer_head.py:389-470: one (cls, bbox) loss pair per decoder layer, the last one under the plain names.

        Same values as ``multi_apply(self.loss_single, ...)`` (the reference; kept above, and tested equal), organised for the
        device: the matching costs of ALL decoder layers and samples are computed first and read back in ONE transfer, the
        assignments are solved on the host, and targets / losses are then built from the host-side matches without a single
        further read-back (the reference syncs ~8 times per layer and sample: cost matrix, nonzero / unique of the sampler,
        boolean-mask indexing, .item() of the normaliser)."""
        import numpy as np
        from scipy.optimize import linear_sum_assignment
        assert gt_bboxes_ignore is None
        all_cls, all_box = preds_dicts["all_cls_scores"], preds_dicts["all_bbox_preds"]
        assert preds_dicts.get("enc_cls_scores") is None
        n_layers, bs, nq = all_cls.shape[0], all_cls.shape[1], all_cls.shape[2]
        device = all_cls.device
        gt_boxes = [torch.cat((b.gravity_center, b.tensor[:, 3:]), dim=1).to(device) for b in gt_bboxes_list]
        gt_labels = [l.to(device) for l in gt_labels_list]
        with torch.no_grad():
            blocks = [self.assigner.cost_matrix(all_box[lv, i], all_cls[lv, i], gt_boxes[i], gt_labels[i]).reshape(-1)
                      for lv in range(n_layers) for i in range(bs) if gt_boxes[i].size(0) > 0]
            host = torch.cat(blocks).cpu().numpy() if blocks else np.zeros((0,), np.float32)      # the one read-back
        out, off = {}, 0
        distributed = dist.is_available() and dist.is_initialized()
        tdims = None
        for lv in range(n_layers):
            q_idx, g_idx, g_smp = [], [], []
            for i in range(bs):
                m = gt_boxes[i].size(0)
                if m == 0:
                    continue
                rows, cols = linear_sum_assignment(host[off:off + nq * m].reshape(nq, m))
                off += nq * m
                q_idx.append(rows + i * nq)
                g_idx.append(cols)
                g_smp.append(i)
            num_pos = int(sum(len(r) for r in q_idx))
            num_neg = bs * nq - num_pos
            cls_scores = all_cls[lv].reshape(-1, self.cls_out_channels)
            bbox_preds = all_box[lv].reshape(-1, all_box.size(-1))
            labels = torch.full((bs * nq,), self.num_classes, dtype=torch.long, device=device)
            if num_pos:
                pos = L.h2d(torch.from_numpy(np.concatenate(q_idx)), device)
                sel = [L.h2d(torch.from_numpy(c), device) for c in g_idx]
                labels[pos] = torch.cat([gt_labels[i][c] for i, c in zip(g_smp, sel)])
                pos_boxes = torch.cat([gt_boxes[i][c] for i, c in zip(g_smp, sel)])
            cls_avg_factor = num_pos * 1.0 + num_neg * self.bg_cls_weight
            if self.sync_cls_avg_factor and distributed:
                cls_avg_factor = reduce_mean(cls_scores.new_tensor([cls_avg_factor]))
            cls_avg_factor = max(cls_avg_factor, 1)
            loss_cls = self.loss_cls(cls_scores, labels, cls_scores.new_ones(bs * nq), avg_factor=cls_avg_factor)
            npos = torch.clamp(reduce_mean(loss_cls.new_tensor([num_pos])), min=1).item() if distributed else max(float(num_pos), 1.0)
            if num_pos:
                preds = bbox_preds[pos]
                tdims = self._target_dims(pos_boxes)
                targets = normalize_bbox(pos_boxes[:, :tdims], self.pc_range)
                finite = torch.isfinite(targets).all(dim=-1, keepdim=True)           # (a degenerate box has log(0) sizes)
                targets = torch.where(finite, targets, preds.detach()[:, :targets.size(-1)])
                weights = finite.to(preds.dtype).expand_as(preds)
            else:
                preds = targets = weights = bbox_preds[:0]
            lc, lb = self._box_loss(loss_cls, preds, targets, weights, npos)
            if lv == n_layers - 1:
                out["loss_cls"], out["loss_bbox"] = lc, lb
            else:
                out[f"d{lv}.loss_cls"], out[f"d{lv}.loss_bbox"] = lc, lb
        return {k: out[k] for k in ["loss_cls", "loss_bbox"] + [f"d{i}.{n}" for i in range(n_layers - 1) for n in ("loss_cls", "loss_bbox")]}

    def loss_reference_order(self, gt_bboxes_list, gt_labels_list, preds_dicts, gt_bboxes_ignore=None, img_metas=None):
        """The reference's own organisation (multi_apply over loss_single): the definition ``loss`` is tested against."""
        assert gt_bboxes_ignore is None
        all_cls, all_box = preds_dicts["all_cls_scores"], preds_dicts["all_bbox_preds"]
        n = len(all_cls)
        device = gt_labels_list[0].device
        gt_bboxes_list = [torch.cat((b.gravity_center, b.tensor[:, 3:]), dim=1).to(device) for b in gt_bboxes_list]
        losses_cls, losses_bbox = multi_apply(self.loss_single, all_cls, all_box, [gt_bboxes_list] * n, [gt_labels_list] * n,
                                              [gt_bboxes_ignore] * n)
        out = {"loss_cls": losses_cls[-1], "loss_bbox": losses_bbox[-1]}
        for i, (lc, lb) in enumerate(zip(losses_cls[:-1], losses_bbox[:-1])):
            out[f"d{i}.loss_cls"], out[f"d{i}.loss_bbox"] = lc, lb
        return out

    def get_bboxes(self, preds_dicts, img_metas, rescale=False):
        """bevformer_head.py:473-520: decoded boxes (gravity centre -> bottom centre), scores, labels per sample"""
        out = []
        for i, preds in enumerate(self.bbox_coder.decode(preds_dicts)):
            bboxes = preds["bboxes"]
            bboxes[:, 2] = bboxes[:, 2] - bboxes[:, 5] * 0.5
            out.append([img_metas[i]["box_type_3d"](bboxes, self._box_dims(bboxes)), preds["scores"], preds["labels"]])
        return out


@MODELS.register_module()
class BEVFormerHead(_SetPredictionHead):
    """bevformer_head.py:21-520"""

    def __init__(self, *args, code_weights=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.real_w = self.pc_range[3] - self.pc_range[0]
        self.real_h = self.pc_range[4] - self.pc_range[1]
        cw = code_weights if code_weights is not None else [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2]
        self.code_weights = nn.Parameter(torch.tensor(cw, requires_grad=False), requires_grad=False)

    def forward(self, mlvl_feats, img_metas, prev_bev=None, only_bev=False):
        """:122-207  mlvl_feats: per level [bs, num_cam, C, H, W]"""
        bs = mlvl_feats[0].shape[0]
        dtype = mlvl_feats[0].dtype
        object_query_embeds = self.query_embedding.weight.to(dtype)
        bev_queries = self.bev_embedding.weight.to(dtype)
        bev_mask = torch.zeros((bs, self.bev_h, self.bev_w), device=bev_queries.device).to(dtype)
        bev_pos = self.positional_encoding(bev_mask).to(dtype)
        grid_length = (self.real_h / self.bev_h, self.real_w / self.bev_w)
        if only_bev:
            return self.transformer.get_bev_features(mlvl_feats, bev_queries, self.bev_h, self.bev_w, grid_length=grid_length,
                                                     bev_pos=bev_pos, img_metas=img_metas, prev_bev=prev_bev)
        bev_embed, hs, init_reference, inter_references = self.transformer(
            mlvl_feats, bev_queries, object_query_embeds, self.bev_h, self.bev_w, grid_length=grid_length, bev_pos=bev_pos,
            reg_branches=self.reg_branches if self.with_box_refine else None, cls_branches=None, img_metas=img_metas,
            prev_bev=prev_bev)
        hs = hs.permute(0, 2, 1, 3)
        classes, coords = self._decode_layers(hs, init_reference, inter_references)
        return {"bev_embed": bev_embed.permute(1, 0, 2), "all_cls_scores": classes, "all_bbox_preds": coords,
                "enc_cls_scores": None, "enc_bbox_preds": None, "hs": hs, "query_embed": object_query_embeds}

    def _target_dims(self, gt_bboxes):
        return gt_bboxes.shape[-1]

    def _box_dims(self, bboxes):
        return bboxes.shape[-1]

    def _box_loss(self, loss_cls, bbox_preds, targets, box_w, num_pos):
        """:375-386 on the rows with finite targets"""
        box_w = box_w * self.code_weights
        loss_bbox = self.loss_bbox(bbox_preds[:, :10], targets[:, :10], box_w[:, :10], avg_factor=num_pos)
        return torch.nan_to_num(loss_cls), torch.nan_to_num(loss_bbox)


@MODELS.register_module()
class DGCNN3DHead(_SetPredictionHead):
    """dgcnn3d_head.py:17-510 (LidarFormer / MVPFormer teachers)"""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.voxel_size = self.bbox_coder.voxel_size
        self.bev_shape = (int((self.pc_range[3] - self.pc_range[0]) / self.voxel_size[0]),
                          int((self.pc_range[4] - self.pc_range[1]) / self.voxel_size[1]))

    def forward(self, mlvl_feats, only_bev=False):
        """:87-169  mlvl_feats: per level [bs, C, H, W]; only_bev: stop after the BEV encoder -> {'bev_embed': ...}"""
        bs = mlvl_feats[0].size(0)
        img_masks = mlvl_feats[0].new_zeros((bs,) + self.bev_shape)
        mlvl_masks = [F.interpolate(img_masks[None], size=f.shape[-2:]).to(torch.bool).squeeze(0) for f in mlvl_feats]
        mlvl_pos = [self.positional_encoding(m) for m in mlvl_masks]
        query_embeds = self.query_embedding.weight
        bev_queries = self.bev_embedding.weight
        bev_mask = torch.zeros((bs, self.bev_h, self.bev_w), device=bev_queries.device)
        bev_pos = self.positional_encoding(bev_mask)
        hs, init_reference, inter_references, bev_embed, _, _ = self.transformer(
            mlvl_feats, bev_queries, mlvl_masks, bev_mask, bev_pos, query_embeds, mlvl_pos,
            reg_branches=self.reg_branches if self.with_box_refine else None, cls_branches=None, only_bev=only_bev)
        if only_bev:
            return {"bev_embed": bev_embed}
        hs = hs.permute(0, 2, 1, 3)
        classes, coords = self._decode_layers(hs, init_reference, inter_references)
        return {"all_cls_scores": classes, "all_bbox_preds": coords, "enc_cls_scores": None, "enc_bbox_preds": None,
                "bev_embed": bev_embed, "hs": hs, "query_embed": query_embeds}

    def _get_target_single(self, cls_score, bbox_pred, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
        """:171-197 takes (boxes, labels) in the other order"""
        return super()._get_target_single(cls_score, bbox_pred, gt_labels, gt_bboxes, gt_bboxes_ignore)

    def get_targets(self, cls_scores_list, bbox_preds_list, gt_bboxes_list, gt_labels_list, gt_bboxes_ignore_list=None):
        assert gt_bboxes_ignore_list is None
        ignore = [None for _ in cls_scores_list]
        labels, label_w, box_t, box_w, pos, neg = multi_apply(self._get_target_single, cls_scores_list, bbox_preds_list,
                                                              gt_bboxes_list, gt_labels_list, ignore)
        return labels, label_w, box_t, box_w, sum(i.numel() for i in pos), sum(i.numel() for i in neg)

    def _target_dims(self, gt_bboxes):
        return self.code_size - 1

    def _box_dims(self, bboxes):
        return 9 if bboxes.size(-1) == 9 else 7

    def _box_loss(self, loss_cls, bbox_preds, targets, box_w, num_pos):
        """:247-254 on the rows with finite targets"""
        loss_bbox = self.loss_bbox(bbox_preds[:, :8], targets[:, :8], box_w[:, :8], avg_factor=num_pos)
        if self.code_size > 8:
            loss_bbox = loss_bbox + 0.2 * self.loss_bbox(bbox_preds[:, 8:], targets[:, 8:], box_w[:, 8:], avg_factor=num_pos)
        return loss_cls, loss_bbox
