"""BEVFormer student and its LiDAR -> camera distillation (BASELINE configs[4]):

  mmdet3d/models/detectors/bevformer.py            BEVFormer :16-290 (extract_img_feat :65-93, obtain_history_bev :150-168,
                                                   forward_train :170-229, forward_test / simple_test :231-290)
  mmdet3d/models/detectors/bevformer_distill.py    BEVFormerDistill :91-1020 (inherit :270-291, foreground_scale_mask
                                                   :404-496, fgd_distill_loss :634-812, forward_distill :840-921,
                                                   forward_train :923-987)
  mmdet3d/models/utils/grid_mask.py                GridMask :70-124

The student is ResNet -> FPN -> BEVFormerHead (transformer.py / detr_head.py); the teacher (MVPFormer / LidarFormer,
sparse_encoder.py) gives ``bev_embed`` of its own DGCNN3DHead.  The FGD feature terms are the same kernels as the
BEVDepth recipe (``BEVDepth4DDistill.fgd_distill_loss``: fused 1x1 adaptation + masked MSE, |x|-mean maps, fg-mask
rasteriser -- here with BEVFormer's cell-centre coordinates and fractional out_size_factor).
"""
import copy
import os

import numpy as np
import torch
import torch.nn as nn

from .center_head import LiDARBoxes
from . import _lib as L
from .config import Config
from .detectors import BEVDepth4DDistill, _pick, install_fgd_modules, load_checkpoint
from .distill_loss import ForegroundMaskRasterizer
from .registry import MODELS, build_backbone, build_detector, build_head, build_neck
from . import detr_head, transformer  # noqa: F401  (registers the head / transformer types the configs name)


class GridMask(nn.Module):
    """grid_mask.py:70-124: with probability ``prob`` (training only) zero a regular grid of bands (period d drawn in
    [2, h), band width ~ ratio * d, random phase) in every image plane.  Same numpy draws, in the same order, as the
    reference; the mask is built vectorised on the host and uploaded once."""

    def __init__(self, use_h, use_w, rotate=1, offset=False, ratio=0.5, mode=0, prob=1.0):
        super().__init__()
        self.use_h, self.use_w, self.rotate, self.offset, self.ratio, self.mode = use_h, use_w, rotate, offset, ratio, mode
        self.st_prob = self.prob = prob

    def set_prob(self, epoch, max_epoch):
        self.prob = self.st_prob * epoch / max_epoch

    def forward(self, x):
        if np.random.rand() > self.prob or not self.training:
            return x
        n, c, h, w = x.size()
        hh, ww = int(1.5 * h), int(1.5 * w)
        d = np.random.randint(2, h)
        self.l = min(max(int(d * self.ratio + 0.5), 1), d - 1)
        st_h, st_w = np.random.randint(d), np.random.randint(d)
        r = np.random.randint(self.rotate)
        assert r == 0, "BEVFormer builds GridMask(rotate=1): the band pattern is never rotated"

        def bands(size, start):            # rows s .. min(s + l, size) for s = d * i + start, i < size // d
            k = np.arange(size) - start
            return (k >= 0) & (k % d < self.l) & (k // d < size // d)
        mask = np.ones((hh, ww), np.float32)
        if self.use_h:
            mask[bands(hh, st_h), :] = 0
        if self.use_w:
            mask[:, bands(ww, st_w)] = 0
        mask = mask[(hh - h) // 2:(hh - h) // 2 + h, (ww - w) // 2:(ww - w) // 2 + w]
        mask = L.h2d(torch.from_numpy(np.ascontiguousarray(mask)).to(x.dtype), x.device)
        if self.mode == 1:
            mask = 1 - mask
        if self.offset:
            off = L.h2d(torch.from_numpy(2 * (np.random.rand(h, w) - 0.5)).to(x.dtype), x.device)
            return x * mask + off * (1 - mask)
        return x * mask


@MODELS.register_module()
class BEVFormer(nn.Module):
    """bevformer.py:16-290 (MVXTwoStageDetector with only the image branch and the BEV head built)"""

    def __init__(self, use_grid_mask=False, pts_voxel_layer=None, pts_voxel_encoder=None, pts_middle_encoder=None,
                 pts_fusion_layer=None, img_backbone=None, pts_backbone=None, img_neck=None, pts_neck=None, pts_bbox_head=None,
                 img_roi_head=None, img_rpn_head=None, train_cfg=None, test_cfg=None, pretrained=None, video_test_mode=False,
                 init_cfg=None):
        super().__init__()
        assert pts_voxel_encoder is None and pts_middle_encoder is None and pts_backbone is None, \
            "BEVFormer is a camera-only detector: the point branch of MVXTwoStageDetector is not configured"
        if pts_bbox_head:
            head = dict(pts_bbox_head)
            head.update(train_cfg=train_cfg["pts"] if train_cfg else None, test_cfg=test_cfg["pts"] if test_cfg else None)
            self.pts_bbox_head = build_head(head)
        if img_backbone:
            self.img_backbone = build_backbone(img_backbone)
        if img_neck is not None:
            self.img_neck = build_neck(img_neck)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.grid_mask = GridMask(True, True, rotate=1, offset=False, ratio=0.5, mode=1, prob=0.7)
        self.use_grid_mask = use_grid_mask
        self.video_test_mode = video_test_mode
        self.prev_frame_info = {"prev_bev": None, "scene_token": None, "prev_pos": 0, "prev_angle": 0}

    with_img_neck = property(lambda self: hasattr(self, "img_neck") and self.img_neck is not None)

    def init_weights(self):
        for name in ("img_backbone", "img_neck", "pts_bbox_head"):
            mod = getattr(self, name, None)
            if mod is not None and hasattr(mod, "init_weights"):
                mod.init_weights()

    def extract_img_feat(self, img, img_metas=None, len_queue=None):
        """:65-93 -> per FPN level [B, num_cams, C, H, W] (or [B / len_queue, len_queue, num_cams, C, H, W])"""
        if img is None:
            return None
        B = img.size(0)
        if img.dim() == 5 and img.size(0) == 1:
            img = img.squeeze(0)
        elif img.dim() == 5 and img.size(0) > 1:
            B, N, C, H, W = img.size()
            img = img.reshape(B * N, C, H, W)
        if self.use_grid_mask:
            img = self.grid_mask(img)
        feats = self.img_backbone(img)
        if isinstance(feats, dict):
            feats = list(feats.values())
        if self.with_img_neck:
            feats = self.img_neck(feats)
        out = []
        for f in feats:
            BN, C, H, W = f.size()
            if len_queue is not None:
                out.append(f.view(int(B / len_queue), len_queue, int(BN / B), C, H, W))
            else:
                out.append(f.view(B, int(BN / B), C, H, W))
        return out

    def extract_feat(self, img, img_metas=None, len_queue=None):
        return self.extract_img_feat(img, img_metas, len_queue=len_queue)

    def forward_pts_train(self, pts_feats, gt_bboxes_3d, gt_labels_3d, img_metas, gt_bboxes_ignore=None, prev_bev=None,
                          get_preds=False):
        outs = self.pts_bbox_head(pts_feats, img_metas, prev_bev)
        losses = self.pts_bbox_head.loss(gt_bboxes_3d, gt_labels_3d, outs, img_metas=img_metas)
        return (outs, losses) if get_preds else losses

    def forward(self, return_loss=True, **kwargs):
        return self.forward_train(**kwargs) if return_loss else self.forward_test(**kwargs)

    def obtain_history_bev(self, imgs_queue, img_metas_list):
        """:150-168: the BEV map of the history frames, frame by frame, in eval mode and without gradients"""
        self.eval()
        with torch.no_grad():
            prev_bev = None
            bs, len_queue, num_cams, C, H, W = imgs_queue.shape
            feats = self.extract_feat(img=imgs_queue.reshape(bs * len_queue, num_cams, C, H, W), len_queue=len_queue)
            for i in range(len_queue):
                metas = [each[i] for each in img_metas_list]
                if not metas[0]["prev_bev_exists"]:
                    prev_bev = None
                prev_bev = self.pts_bbox_head([lvl[:, i] for lvl in feats], metas, prev_bev, only_bev=True)
        self.train()                                     # the reference switches back to train() unconditionally
        return prev_bev

    def _current_frame(self, img, img_metas):
        """shared head of forward_train (:204-216): history BEV, then the features of the last frame of the queue"""
        len_queue = img.size(1)
        prev_img, img = img[:, :-1, ...], img[:, -1, ...]
        prev_bev = self.obtain_history_bev(prev_img, copy.deepcopy(img_metas)) if len_queue > 1 else None
        img_metas = [each[len_queue - 1] for each in img_metas]
        if not img_metas[0]["prev_bev_exists"]:
            prev_bev = None
        return self.extract_feat(img=img, img_metas=img_metas), img_metas, prev_bev

    def forward_train(self, points=None, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None, gt_labels=None, gt_bboxes=None,
                      img=None, proposals=None, gt_bboxes_ignore=None, img_depth=None, img_mask=None):
        """:170-229  img [bs, queue, num_cams, 3, H, W]; img_metas: per sample {queue index: meta dict}"""
        img_feats, img_metas, prev_bev = self._current_frame(img, img_metas)
        return dict(self.forward_pts_train(img_feats, gt_bboxes_3d, gt_labels_3d, img_metas, gt_bboxes_ignore, prev_bev))

    def forward_test(self, img_metas, img=None, **kwargs):
        """:231-268: single-sample streaming inference carrying the previous BEV map and ego pose"""
        assert isinstance(img_metas, list)
        img = [img] if img is None else img
        meta = img_metas[0][0]
        info = self.prev_frame_info
        if meta["scene_token"] != info["scene_token"] or not self.video_test_mode:
            info["prev_bev"] = None
        info["scene_token"] = meta["scene_token"]
        pos, angle = copy.deepcopy(meta["can_bus"][:3]), copy.deepcopy(meta["can_bus"][-1])
        if info["prev_bev"] is not None:
            meta["can_bus"][:3] -= info["prev_pos"]
            meta["can_bus"][-1] -= info["prev_angle"]
        else:
            meta["can_bus"][-1] = 0
            meta["can_bus"][:3] = 0
        new_prev_bev, results = self.simple_test(img_metas[0], img[0], prev_bev=info["prev_bev"], **kwargs)
        info["prev_pos"], info["prev_angle"], info["prev_bev"] = pos, angle, new_prev_bev
        return results

    def simple_test(self, img_metas, img=None, prev_bev=None, rescale=False):
        """:270-290 -> (bev_embed, [dict(pts_bbox=dict(boxes_3d, scores_3d, labels_3d))])"""
        feats = self.extract_feat(img=img, img_metas=img_metas)
        outs = self.pts_bbox_head(feats, img_metas, prev_bev=prev_bev)
        results = [dict(pts_bbox=dict(boxes_3d=b, scores_3d=s.cpu(), labels_3d=l.cpu()))
                   for b, s, l in self.pts_bbox_head.get_bboxes(outs, img_metas, rescale=rescale)]
        return outs["bev_embed"], results


@MODELS.register_module()
class BEVFormerDistill(BEVFormer):
    """bevformer_distill.py:91-1020"""

    fused_adapt_mse = True
    fgd_distill_loss = BEVDepth4DDistill.fgd_distill_loss        # the same FGD arithmetic and kernels (see module docstring)

    def __init__(self, teacher_config, teacher_ckpt, distill_type, distill_params, eval_teacher=True, self_ckpt=None,
                 inherit_head=False, inherit_decoder=False, inherit_query=False, no_bg=False, img_norm_cfg=None, config_root=None,
                 **kwargs):
        super().__init__(**kwargs)
        self.img_norm_cfg, self.eval_teacher, self.no_bg = img_norm_cfg, eval_teacher, no_bg
        if isinstance(teacher_config, str):
            path = teacher_config
            if not os.path.isabs(path) and not os.path.exists(path) and config_root:
                path = os.path.join(config_root, path)
            teacher_config = Config.fromfile(path)
        self.teacher_model = build_detector(teacher_config["model"] if "model" in teacher_config else teacher_config)
        has_ckpt = isinstance(teacher_ckpt, str) and teacher_ckpt.lower() != "none"
        if has_ckpt:
            load_checkpoint(self.teacher_model, teacher_ckpt, what="teacher")                     # :102-106 (strict)
        for p in self.teacher_model.parameters():
            p.requires_grad_(False)
        self.inherit_head, self.inherit_decoder, self.inherit_query = inherit_head, inherit_decoder, inherit_query
        for flag in (inherit_head, inherit_decoder, inherit_query):                                # :115-120
            assert not flag or has_ckpt, "inheriting the TRAINED teacher head / decoder / queries needs teacher_ckpt"
        assert distill_type in ["fgd"]
        self.distill_type = distill_type
        dp = self.distill_params = distill_params
        install_fgd_modules(self, dp)
        if isinstance(self_ckpt, str) and self_ckpt.lower() != "none":
            load_checkpoint(self, self_ckpt, what="student", allow_missing=True)
        tcfg = self.pts_bbox_head.train_cfg
        self._fg_raster = ForegroundMaskRasterizer(tcfg["grid_size"], tcfg["point_cloud_range"], tcfg["voxel_size"],
                                                   cell_center=True)
        self._epoch = 0
        self.iter = 0

    # ---- the teacher is a plain attribute: hidden from parameters() / state_dict() / DDP (:989-1020) ----
    def __setattr__(self, name, value):
        if name == "teacher_model":
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    def _apply(self, fn, *args, **kwargs):
        self.teacher_model._apply(fn)
        return super()._apply(fn, *args, **kwargs)

    def train(self, mode=True):
        self.teacher_model.train(False if self.eval_teacher else mode)
        return super().train(mode)

    def set_epoch(self, epoch):
        self._epoch = epoch

    def init_weights(self):
        super().init_weights()
        self.inherit()

    def inherit(self):
        """:270-291"""
        if not self.inherit_head:
            return
        s, t = self.pts_bbox_head, self.teacher_model.pts_bbox_head
        s.cls_branches.load_state_dict(t.cls_branches.state_dict(), strict=False)
        s.reg_branches.load_state_dict(t.reg_branches.state_dict(), strict=False)
        if self.inherit_decoder:
            s.transformer.decoder.load_state_dict(t.transformer.decoder.state_dict(), strict=False)
        if self.inherit_query:
            s.query_embedding.load_state_dict(t.query_embedding.state_dict(), strict=False)

    def add_fp_as_fg(self, mode, fg_mask, heatmaps, teacher_preds, student_preds):
        """add_fp_as_fg_bbox (:555-631): cells inside a confident TEACHER box (score > output_threshold) and outside every
        ground-truth box.  Returns (fp mask, 1 / count scale, count) like the BEVDepth variant."""
        dp = self.distill_params
        thres = dp["output_threshold"]
        B, _, H, W = fg_mask.shape
        boxes = [b.tensor[(s > thres).cpu()] if len(s) else b.tensor[:0] for b, s, _ in teacher_preds]
        inside, _, _ = self._fg_raster(H, W, [b.cpu() for b in boxes], fg_mask.device)
        # the reference lays this mask out [x, y] -- it reshapes the x-major cell list without the transpose that
        # foreground_scale_mask applies (:486-489 vs :621) -- and multiplies it with the [y, x] attention map as is: kept
        fp = ((inside > 0) & (fg_mask == 0)).float().transpose(2, 3).contiguous()
        n_fp = fp.sum(dim=(1, 2, 3))
        return fp, fp / n_fp.clamp(min=1).view(-1, 1, 1, 1), n_fp

    def forward_distill(self, points, img_metas, gt_bboxes_3d, gt_labels_3d, img_feats, lss_feat, bev_backbone_feats,
                        student_query, hs_feats, preds, heatmaps, img_inputs=None):
        """:840-921  img_feats = the student's bev_embed [bs, H*W, C]"""
        dp = self.distill_params
        with torch.no_grad():
            _, teacher_x = self.teacher_model.extract_feat(points, None, img_metas)
            # the teacher's decoder and box decoding only feed the false-positive term: skipped when no position uses it
            needs_boxes = any(m != "none" for m in dp["fp_as_foreground"]) and self._epoch >= dp["fp_epoch"]
            teacher_outs = self.teacher_model.pts_bbox_head(teacher_x, only_bev=not needs_boxes)
            teacher_bev = teacher_outs["bev_embed"]
            teacher_preds = self.teacher_model.pts_bbox_head.get_bboxes(teacher_outs, img_metas, rescale=False) if needs_boxes else None
        assert len(set(dp["student_feat_pos"])) == len(dp["student_feat_pos"]) == len(dp["teacher_feat_pos"])
        out = {}
        for index, (spos, tpos) in enumerate(zip(dp["student_feat_pos"], dp["teacher_feat_pos"])):
            assert spos == "head" and tpos == "head", "the shipped BEVFormer recipes distil the BEV embedding ('head') only"
            s, t = img_feats.permute(0, 2, 1), teacher_bev.permute(0, 2, 1)
            sH, tH = int(s.shape[2] ** 0.5), int(t.shape[2] ** 0.5)
            # [bs, C, H, W] views of the [bs, H*W, C] embeddings == channels-last tensors: no copy, and the layout the
            # fused adaptation / masked-MSE kernels read
            s = s.reshape(s.shape[0], s.shape[1], sH, sH)
            t = t.reshape(t.shape[0], t.shape[1], tH, tH)
            losses = self.fgd_distill_loss(t, s, gt_bboxes_3d, gt_labels_3d, None, heatmaps, teacher_preds, preds, index)
            if self.no_bg:
                losses.pop("kd_bg_feat_loss", None)
            for k, v in losses.items():
                out[f"{k}_{spos}_{tpos}"] = v
        return out

    def forward_train(self, points=None, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None, gt_labels=None, gt_bboxes=None,
                      img=None, proposals=None, gt_bboxes_ignore=None):
        """:923-987"""
        img_feats, img_metas, prev_bev = self._current_frame(img, img_metas)
        outs, losses_pts = self.forward_pts_train(img_feats, gt_bboxes_3d, gt_labels_3d, img_metas, gt_bboxes_ignore, prev_bev,
                                                  get_preds=True)
        losses = dict(losses_pts)
        losses.update(self.forward_distill(points, img_metas, gt_bboxes_3d, gt_labels_3d, outs["bev_embed"], None, None,
                                           outs["query_embed"], outs["hs"], outs["all_bbox_preds"], None))
        self.iter += 1
        return losses


def make_bevformer_batch(B, rng, device, queue_length=4, n_cams=6, img_size=(928, 1600), n_points=(200000, 50000, 150000),
                         n_boxes=30):
    """One synthetic batch in the input contract of the BEVFormer distillation configs (CustomNuScenesDataset with
    ``queue_length`` frames, custom_nus-3d.py): img f32[B, queue, num_cams, 3, H, W]; img_metas per sample {frame: dict(
    can_bus[18], lidar2img [num_cams x 4x4], img_shape, prev_bev_exists, scene_token, box_type_3d)}; MVP virtual-point clouds
    f32[n, 17] (n_points = real, painted, virtual); 9-dof boxes; labels."""
    from . import synthetic as syn
    H, W = img_size
    g = torch.Generator(device="cpu").manual_seed(int(rng.integers(0, 2 ** 31)))
    img = torch.randn((B, queue_length, n_cams, 3, H, W), generator=g).to(device)
    rig = syn.camera_rig(B, rng, n_cams=n_cams, input_size=(900, 1600))
    metas, points, boxes, labels = [], [], [], []
    for b in range(B):
        l2i = []
        for n in range(n_cams):                     # lidar -> camera -> pixel: K [R^T | -R^T t]
            R, t, K = rig["rots"][b, n].astype(np.float64), rig["trans"][b, n].astype(np.float64), rig["intrins"][b, n].astype(np.float64)
            ext = np.eye(4)
            ext[:3, :3], ext[:3, 3] = R.T, -R.T @ t
            Kh = np.eye(4)
            Kh[:3, :3] = K
            Kh[:2] *= W / 1600.0                        # the rig's intrinsics are those of a 1600-pixel-wide image
            l2i.append(Kh @ ext)
        frames = {}
        for q in range(queue_length):
            can_bus = np.zeros(18)
            can_bus[:3] = [rng.uniform(0.2, 1.0), rng.uniform(-0.1, 0.1), 0.0] if q else 0.0      # ego translation since frame q-1
            can_bus[-2] = rng.uniform(-0.05, 0.05)                                                 # ego yaw (rad)
            can_bus[-1] = rng.uniform(-2.0, 2.0) if q else 0.0                                     # yaw change (deg)
            frames[q] = dict(can_bus=can_bus, lidar2img=[m.copy() for m in l2i], img_shape=[(H, W, 3)] * n_cams,
                             prev_bev_exists=q > 0, scene_token=f"scene-{b}", box_type_3d=lambda t, box_dim=9: LiDARBoxes(t))
        metas.append(frames)
        points.append(torch.from_numpy(syn.mvp_virtual_points(*n_points, rng)).to(device))
        bx, lb = syn.gt_boxes(n_boxes, rng)
        boxes.append(LiDARBoxes(bx))
        labels.append(torch.from_numpy(lb).to(device))
    return dict(img=img, img_metas=metas, points=points, gt_bboxes_3d=boxes, gt_labels_3d=labels)
