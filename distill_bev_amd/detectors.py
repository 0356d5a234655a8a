"""Detectors on the distillation hot path -- registry mirror of
  mmdet3d/models/detectors/centerpoint.py            CenterPoint.extract_pts_feat :40-70, forward_pts_train :72-104
  mmdet3d/models/detectors/dynamic_centerpoint.py    DynamicCenterPoint :13-93
  mmdet3d/models/detectors/mvx_two_stage.py          voxelize :217-242 (hard voxels)
  mmdet3d/models/detectors/bevdet.py                 BEVDet.image_encoder :21-35
  mmdet3d/models/detectors/bevdet_distill.py         BEVDetDistill :155-373, fgd_distill_loss :973-1324,
                                                     add_fp_as_fg :846-970, forward_distill :1411-1507
  mmdet3d/models/detectors/bevdet_distill_more.py    BEVDet4DDistill.shift_feature :41-94,
                                                     BEVDepth4DDistill :334-522
Attribute names (= checkpoint keys) and the loss-dict keys are the reference's.  The hot ops go to
the gfx950 library: fused lift-splat, voxelize / dynamic scatter / pillars scatter, fg-mask
rasteriser, |x|-mean maps, fused masked MSE.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from .center_head import LiDARBoxes, clip_sigmoid  # noqa: F401
from .config import Config
from .bn_act import bn_act
from .distill_loss import (ForegroundMaskRasterizer, UpsampleBilinearAC, fgd_feature_losses, fgd_feature_losses_fused_adapt,
                           fused_adapt_eligible)
from .registry import MODELS, build_backbone, build_detector, build_head, build_loss, build_neck
from .voxel import Voxelization

# make every registered component importable through this module
from . import lss as LSS  # noqa: E402
from . import nets, pillar_encoder, pillars, view_transformer  # noqa: F401,E402
from . import msda, sparse_encoder, spconv  # noqa: F401,E402  (voxel teachers / deformable attention of the 'next' rows)

MODELS.register_module(name="PointPillarsScatter", module=pillars.PointPillarsScatter)


@MODELS.register_module()
class CenterPoint(nn.Module):
    """LiDAR teacher (pillar variant): voxelize -> pillar VFE -> scatter -> SECOND -> SECONDFPN -> CenterHead."""

    def __init__(self, pts_voxel_layer=None, pts_voxel_encoder=None, pts_middle_encoder=None,
                 pts_fusion_layer=None, img_backbone=None, pts_backbone=None, img_neck=None, pts_neck=None,
                 pts_bbox_head=None, img_roi_head=None, img_rpn_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, init_cfg=None, test_dist2velo=False, lidar_interval=1.0 / 20):
        super().__init__()
        if pts_voxel_layer:
            self.pts_voxel_layer = Voxelization(**pts_voxel_layer)
        if pts_voxel_encoder:
            self.pts_voxel_encoder = MODELS.build(pts_voxel_encoder)
        if pts_middle_encoder:
            self.pts_middle_encoder = MODELS.build(pts_middle_encoder)
        if pts_backbone:
            self.pts_backbone = build_backbone(pts_backbone)
        if pts_neck is not None:
            self.pts_neck = build_neck(pts_neck)
        if pts_bbox_head:
            pts_train_cfg = train_cfg["pts"] if train_cfg else None
            pts_test_cfg = test_cfg["pts"] if test_cfg else None
            head = dict(pts_bbox_head)
            head.update(train_cfg=pts_train_cfg, test_cfg=pts_test_cfg)
            self.pts_bbox_head = build_head(head)
        if img_backbone:
            self.img_backbone = build_backbone(img_backbone)
        if img_neck is not None:
            self.img_neck = build_neck(img_neck)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    with_pts_bbox = property(lambda self: hasattr(self, "pts_bbox_head") and self.pts_bbox_head is not None)
    with_pts_neck = property(lambda self: hasattr(self, "pts_neck") and self.pts_neck is not None)
    with_img_neck = property(lambda self: hasattr(self, "img_neck") and self.img_neck is not None)

    @torch.no_grad()
    def voxelize(self, points):
        """mvx_two_stage.py:217-242 (hard voxelization, per sample, batch-padded coors)."""
        voxels, coors, num_points = [], [], []
        for res in points:
            v, c, n = self.pts_voxel_layer(res)
            voxels.append(v); coors.append(c); num_points.append(n)
        voxels = torch.cat(voxels, dim=0)
        num_points = torch.cat(num_points, dim=0)
        coors_batch = torch.cat([F.pad(c, (1, 0), mode="constant", value=i) for i, c in enumerate(coors)], dim=0)
        return voxels, num_points, coors_batch

    def extract_pts_feat(self, pts, img_feats=None, img_metas=None, return_canvas=False, return_backbone_feature=False):
        if not self.with_pts_bbox:
            return None
        outputs = []
        voxels, num_points, coors = self.voxelize(pts)
        voxel_features = self.pts_voxel_encoder(voxels, num_points, coors)
        x = self.pts_middle_encoder(voxel_features, coors, len(pts))
        return self._backbone_neck(x, outputs, return_canvas, return_backbone_feature)

    def _backbone_neck(self, x, outputs, return_canvas, return_backbone_feature):
        if return_canvas:
            outputs.append(x)
        x = self.pts_backbone(x)
        if return_backbone_feature:
            outputs.append(x)
        if self.with_pts_neck:
            x = self.pts_neck(x)
            if len(outputs) == 0:
                outputs = x
            else:
                outputs.insert(0, x)
        return outputs

    def forward_pts_train(self, pts_feats, gt_bboxes_3d, gt_labels_3d, img_metas=None, gt_bboxes_ignore=None,
                          get_preds=False, get_targets=False):
        outs = self.pts_bbox_head(pts_feats)
        losses = self.pts_bbox_head.loss(gt_bboxes_3d, gt_labels_3d, outs, get_targets=get_targets)
        return (outs, losses) if get_preds else losses


@MODELS.register_module()
class DynamicCenterPoint(CenterPoint):
    @torch.no_grad()
    def voxelize(self, points):
        """dynamic_centerpoint.py:71-93."""
        coors = [self.pts_voxel_layer(res) for res in points]
        pts = torch.cat(points, dim=0)
        coors_batch = torch.cat([F.pad(c, (1, 0), mode="constant", value=i) for i, c in enumerate(coors)], dim=0)
        return pts, coors_batch

    def extract_pts_feat(self, pts, img_feats=None, img_metas=None, return_canvas=False, return_backbone_feature=False):
        if not self.with_pts_bbox:
            return None
        outputs = []
        if getattr(self, "use_fused_pillar_path", True) and pillar_encoder.fused_pillar_canvas_eligible(
                self.pts_voxel_layer, self.pts_voxel_encoder, self.pts_middle_encoder):
            # frozen teacher: voxelize -> PFN -> max -> canvas in one asynchronous library call
            x = pillar_encoder.fused_pillar_canvas(pts, self.pts_voxel_layer, self.pts_voxel_encoder,
                                                   self.pts_middle_encoder)
            return self._backbone_neck(x, outputs, return_canvas, return_backbone_feature)
        if (getattr(self, "use_fused_pillar_path", True) and not self.training and pts[0].is_cuda
                and isinstance(self.pts_voxel_encoder, pillar_encoder.DynamicPillarFeatureNet)):
            L.note_fallback("pillar_vfe", "frozen pillar teacher outside the fused kernel's recipe")
        voxels, coors = self.voxelize(pts)
        coors = coors.type(torch.int32)
        batch_size = len(pts)
        voxel_features, feature_coors = self.pts_voxel_encoder(voxels, coors, batch_size)
        x = self.pts_middle_encoder(voxel_features, feature_coors.type(torch.int32), batch_size)
        return self._backbone_neck(x, outputs, return_canvas, return_backbone_feature)


# --------------------------------------------------------------------------------------
def _conv_norm_relu(conv, norm, x):
    """relu(norm(conv(x))).  In training mode the norm subtracts the batch mean, so the convolution's bias cancels: the library
    convolution then runs without its separate bias pass and without the bias-gradient reduction (colsum.conv_bn_cancelled_bias)."""
    from .colsum import BiasSumConv2d, cancelled_bias_ready, conv_bn_cancelled_bias
    if type(conv) in (nn.Conv2d, BiasSumConv2d) and cancelled_bias_ready(conv, norm, x):
        return conv_bn_cancelled_bias(conv, norm, x, lambda z, pre=None: bn_act(z, norm, None, True, pre=pre))
    return bn_act(conv(x), norm, None, True)


class ThreeLayer(nn.Module):
    """bevdet_distill.py:99-132: three (conv + BN + ReLU) stages; the first one carries kernel/stride."""

    def __init__(self, in_features, hidden_features=None, out_features=None, kernel_size=4, stride=4, padding=0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.conv1 = nn.Conv2d(in_features, hidden_features, kernel_size=kernel_size, stride=stride, padding=padding)
        self.norm1 = nn.BatchNorm2d(hidden_features); self.act1 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(hidden_features, hidden_features, kernel_size=1)
        self.norm2 = nn.BatchNorm2d(hidden_features); self.act2 = nn.ReLU(inplace=True)
        self.conv3 = nn.Conv2d(hidden_features, out_features, kernel_size=1)
        self.norm3 = nn.BatchNorm2d(out_features); self.act3 = nn.ReLU(inplace=True)

    def forward(self, x):
        x = _conv_norm_relu(self.conv1, self.norm1, x)         # norm -> relu on the fused kernel when eligible
        x = _conv_norm_relu(self.conv2, self.norm2, x)
        return _conv_norm_relu(self.conv3, self.norm3, x)


class TwoLayer(nn.Module):
    """bevdet_distill.py:70-97."""

    def __init__(self, in_features, hidden_features=None, out_features=None, kernel_size=4, stride=4, padding=0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.conv1 = nn.Conv2d(in_features, hidden_features, kernel_size=kernel_size, stride=stride, padding=padding)
        self.norm1 = nn.BatchNorm2d(hidden_features); self.act1 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(hidden_features, out_features, kernel_size=1)
        self.norm2 = nn.BatchNorm2d(out_features); self.act2 = nn.ReLU(inplace=True)

    def forward(self, x):
        return _conv_norm_relu(self.conv2, self.norm2, _conv_norm_relu(self.conv1, self.norm1, x))


def load_checkpoint(module, path, what="model", allow_missing=False):
    """mmcv.runner.load_checkpoint for a local file in the mmdet3d format (dict with 'state_dict' [+ 'meta',
    'optimizer']) or a bare state dict; a DataParallel 'module.' prefix is stripped.  A missing file raises (mmcv
    does too); keys the checkpoint does not provide raise unless allow_missing (mmcv only logs them -- a teacher with
    silently random layers would distil noise); keys the module does not have are reported and ignored."""
    import warnings
    if not os.path.isfile(path):
        raise FileNotFoundError(f"{what} checkpoint '{path}' does not exist")
    ck = torch.load(path, map_location="cpu")
    sd = ck.get("state_dict", ck) if isinstance(ck, dict) else ck
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    missing = [k for k in missing if not k.endswith("num_batches_tracked")]
    if unexpected:
        warnings.warn(f"{what} checkpoint {path}: {len(unexpected)} unexpected keys ignored, e.g. {unexpected[:4]}")
    if missing:
        msg = f"{what} checkpoint {path}: {len(missing)} parameters/buffers not in the file, e.g. {missing[:6]}"
        if not allow_missing:
            raise RuntimeError(msg)
        warnings.warn(msg)
    return missing, unexpected


def _as_list(v, n):
    return list(v) if isinstance(v, (list, tuple)) else [v for _ in range(n)]


def _pick(lst, index):
    return lst[index] if len(lst) > 1 else lst[0]


def install_fgd_modules(self, dp):
    """The trainable pieces of the FGD recipe (bevdet_distill.py:178-357 == bevformer_distill.py:130-262): per feature
    position one student adaptation + one teacher adaptation (+ the 3x3 spatial-attention conv), and the criteria."""
    sc, tc = dp["student_channels"], dp["teacher_channels"]
    n = len(sc)
    assert n == len(tc)
    dp["affinity_mode"] = _as_list(dp["affinity_mode"], n)
    dp["fp_as_foreground"] = _as_list(dp["fp_as_foreground"], n)
    dp["adaptation_type"] = _as_list(dp["adaptation_type"], n)
    dp["teacher_adaptation_type"] = _as_list(dp["teacher_adaptation_type"], n)
    sap = dp["student_adaptation_params"]
    cwa, ta = [], []
    for at, tat, s_c, t_c in zip(dp["adaptation_type"], dp["teacher_adaptation_type"], sc, tc):
        if at == "1x1conv":
            cwa.append(nn.Conv2d(s_c, t_c, kernel_size=1, stride=1, padding=0))
        elif at == "3x3conv":
            cwa.append(nn.Conv2d(s_c, t_c, kernel_size=3, stride=1, padding=1))
        elif at == "identity":
            cwa.append(nn.Identity())
        elif at in ("2layer", "3layer"):
            cls = TwoLayer if at == "2layer" else ThreeLayer
            cwa.append(cls(in_features=s_c, out_features=t_c, kernel_size=sap["kernel_size"], stride=sap["stride"]))
        elif at in ("upsample_2layer", "upsample_3layer"):
            cls = TwoLayer if at == "upsample_2layer" else ThreeLayer
            cwa.append(nn.Sequential(
                UpsampleBilinearAC(sap["upsample_factor"]),     # == nn.Upsample(bilinear, align_corners=True)
                cls(in_features=s_c, out_features=t_c, kernel_size=sap["kernel_size"], stride=sap["stride"])))
        elif at == "upsample_1x1conv":
            cwa.append(nn.Sequential(
                UpsampleBilinearAC(sap["upsample_factor"]),
                nn.Conv2d(s_c, t_c, kernel_size=1, stride=1, padding=0)))
        else:
            raise NotImplementedError(at)
        if tat == "identity":
            ta.append(nn.Identity())
        elif tat == "avgpool":
            ta.append(nn.AvgPool2d(**dp["teacher_adaptation_params"]))
        elif tat == "maxpool":
            ta.append(nn.MaxPool2d(**dp["teacher_adaptation_params"]))
        else:
            raise NotImplementedError(tat)
    self.channel_wise_adaptations = nn.ModuleList(cwa)
    self.teacher_adaptations = nn.ModuleList(ta)
    if dp["spatial_mask"]:
        self.spatial_wise_adaptations = nn.ModuleList(
            [nn.Conv2d(1, 1, kernel_size=3, stride=1, padding=1) for _ in sc])
    self.feat_criterion = build_loss(dp["feat_criterion"])
    self.spatial_criterion = build_loss(dp["spatial_criterion"])
    self.channel_criterion = build_loss(dp["channel_criterion"])
    # the masked-MSE kernels ARE the feature criterion: only MSELoss(reduction='none', weight 1) is fused
    fc = self.feat_criterion
    assert type(fc).__name__ == "MSELoss" and fc.reduction == "none" and fc.loss_weight == 1.0, \
        "feat_criterion must be dict(type='MSELoss', reduction='none'): the FGD feature terms run on the masked-MSE kernels"
    assert getattr(self.spatial_criterion, "reduction", None) == "none", \
        "spatial_criterion must use reduction='none' (fgd_distill_loss sums it, bevdet_distill.py:1275-1277)"


@MODELS.register_module()
class BEVDepth4DDistill(CenterPoint):
    """Camera student (BEVDepth4D, two frames) distilled from a LiDAR teacher with the FGD loss."""

    def __init__(self, img_view_transformer, img_bev_encoder_backbone, img_bev_encoder_neck,
                 teacher_config=None, teacher_ckpt=None, distill_type=None, distill_params=None, eval_teacher=True, self_ckpt=None,
                 inherit_head=False, aligned=False, distill=None, pre_process=None, pre_process_neck=None,
                 detach=True, test_adj_ids=None, before=False, interpolation_mode="bilinear",
                 bevdepth_bev_forward=False, config_root=None, **kwargs):
        super().__init__(**kwargs)
        self.img_view_transformer = build_neck(img_view_transformer)
        self.img_bev_encoder_backbone = build_backbone(img_bev_encoder_backbone)
        self.img_bev_encoder_neck = build_neck(img_bev_encoder_neck)
        assert distill is None and pre_process_neck is None and not bevdepth_bev_forward
        self.aligned, self.detach, self.before, self.interpolation_mode = aligned, detach, before, interpolation_mode
        self.pre_process = pre_process is not None
        if self.pre_process:
            self.pre_process_net = build_backbone(pre_process)
        # ---- teacher (bevdet_distill.py:160-166): a plain attribute, hidden from nn.Module ----
        self.eval_teacher = eval_teacher
        self._epoch = 1
        if teacher_config is None:                 # plain student (BEVDepth4D): no teacher, no distillation modules
            self.teacher_model = None
            self.distill_type, self.distill_params, self.inherit_head = None, None, False
            return
        if isinstance(teacher_config, str):
            path = teacher_config
            if not os.path.isabs(path) and not os.path.exists(path) and config_root:
                path = os.path.join(config_root, path)
            teacher_config = Config.fromfile(path)
        tmodel = teacher_config["model"] if "model" in teacher_config else teacher_config
        self.teacher_model = build_detector(tmodel)
        has_teacher_ckpt = isinstance(teacher_ckpt, str) and teacher_ckpt.lower() != "none"
        if has_teacher_ckpt:
            load_checkpoint(self.teacher_model, teacher_ckpt, what="teacher")       # bevdet_distill.py:163-166
        for p in self.teacher_model.parameters():
            p.requires_grad_(False)
        self._self_ckpt = self_ckpt if isinstance(self_ckpt, str) and self_ckpt.lower() != "none" else None
        self.inherit_head = inherit_head
        if self.inherit_head:                                                        # :175-176
            assert has_teacher_ckpt, "inherit_head=True copies the TRAINED teacher heads: teacher_ckpt is required"
        assert distill_type == "fgd", "only the FGD recipe (the shipped distillation configs) is on the hot path"
        self.distill_type = distill_type
        dp = self.distill_params = distill_params
        install_fgd_modules(self, dp)
        if self._self_ckpt is not None:                                              # :171-173 (after every module exists)
            load_checkpoint(self, self._self_ckpt, what="student", allow_missing=True)
        tcfg = self.pts_bbox_head.train_cfg
        self._fg_raster = ForegroundMaskRasterizer(tcfg["grid_size"], tcfg["point_cloud_range"], tcfg["voxel_size"])
        self._epoch = 1

    # ---- teacher is hidden from parameters()/state_dict()/DDP (bevdet_distill.py:1579-1610) ----
    def __setattr__(self, name, value):
        if name == "teacher_model":
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    def _apply(self, fn, *args, **kwargs):
        if self.teacher_model is not None:
            self.teacher_model._apply(fn)
        return super()._apply(fn, *args, **kwargs)

    def train(self, mode=True):
        if self.teacher_model is not None:
            self.teacher_model.train(False if self.eval_teacher else mode)
        return super().train(mode)

    fused_adapt_mse = True       # class-level switch (tests compare against the unfused conv -> loss kernels sequence)

    def set_epoch(self, epoch):
        self._epoch = epoch

    def init_weights(self):
        for name in ("img_backbone", "img_bev_encoder_backbone"):      # BaseModule.init_weights recursion (pretrained=...)
            mod = getattr(self, name, None)
            if mod is not None and hasattr(mod, "init_weights"):
                mod.init_weights()
        if self.inherit_head:   # bevdet_distill.py:367-373
            self.pts_bbox_head.task_heads.load_state_dict(self.teacher_model.pts_bbox_head.task_heads.state_dict(),
                                                          strict=False)

    # ---- student ----------------------------------------------------------------------------
    def image_encoder(self, img):
        B, N, C, imH, imW = img.shape
        x = img.reshape(B * N, C, imH, imW)
        if getattr(self, "channels_last", False):
            x = x.contiguous(memory_format=torch.channels_last)
        x = self.img_backbone(x)
        if self.with_img_neck:
            x = self.img_neck(x)
            if isinstance(x, (list, tuple)):
                assert len(x) == 1
                x = x[0]
        return x.view(B, N, *x.shape[1:])

    def _feat2bev(self, dev, dt):
        """feature-map index -> BEV metres (bevdet_distill_more.py:70-78) and its inverse, built once per device: the matrix only
        holds the grid constants"""
        cache = self.__dict__.setdefault("_feat2bev_cache", {})
        vt = self.img_view_transformer
        key = (str(dev), dt, vt.dx._version, vt.bx._version, vt.dx.data_ptr(), vt.bx.data_ptr())     # load_state_dict bumps the versions
        if key not in cache:
            cache.clear()
            dx, bx = vt.dx.detach().to(dev, dt), vt.bx.detach().to(dev, dt)
            m = torch.zeros((3, 3), dtype=dt, device=dev)
            m[0, 0] = dx[0]; m[1, 1] = dx[1]
            m[0, 2] = bx[0] - dx[0] / 2.0; m[1, 2] = bx[1] - dx[1] / 2.0
            m[2, 2] = 1
            m = m.view(1, 3, 3)
            cache[key] = (m, LSS.inverse_nosync(m))
        return cache[key]

    def shift_feature(self, input, trans, rots):
        """bevdet_distill_more.py:41-94: warp the adjacent-frame BEV into the current ego frame."""
        n, c, h, w = input.shape
        v = trans[0].shape[1]
        dev, dt = input.device, input.dtype
        xs = torch.linspace(0, w - 1, w, dtype=dt, device=dev).view(1, w).expand(h, w)
        ys = torch.linspace(0, h - 1, h, dtype=dt, device=dev).view(h, 1).expand(h, w)
        grid = torch.stack((xs, ys, torch.ones_like(xs)), -1).view(1, h, w, 3).expand(n, h, w, 3).reshape(n, h, w, 3, 1)
        c02l0 = torch.zeros((n, v, 4, 4), dtype=dt, device=dev)
        c02l0[:, :, :3, :3] = rots[0]; c02l0[:, :, :3, 3] = trans[0]; c02l0[:, :, 3, 3] = 1
        c12l0 = torch.zeros((n, v, 4, 4), dtype=dt, device=dev)
        c12l0[:, :, :3, :3] = rots[1]; c12l0[:, :, :3, 3] = trans[1]; c12l0[:, :, 3, 3] = 1
        l02l1 = c02l0.matmul(LSS.inverse_nosync(c12l0))[:, 0, :, :].view(n, 1, 1, 4, 4)
        # rows / columns (0, 1, 3) as slices: indexing with a Python list uploads the index with a blocking copy, and the host then
        # waits here for everything queued before (82 ms per step)
        l02l1 = torch.cat((l02l1[..., :2, :], l02l1[..., 3:, :]), -2)
        l02l1 = torch.cat((l02l1[..., :2], l02l1[..., 3:]), -1)
        feat2bev, feat2bev_inv = self._feat2bev(dev, dt)
        tf = feat2bev_inv.matmul(l02l1).matmul(feat2bev)
        # tf [n,1,1,3,3] @ grid [n,h,w,3,1]: written as broadcast multiply-adds (a broadcast matmul
        # becomes n*h*w tiny GEMMs -- see lss._apply3x3)
        g = grid[..., 0]
        gx, gy = [tf[..., i, 0] * g[..., 0] + tf[..., i, 1] * g[..., 1] + tf[..., i, 2] * g[..., 2] for i in range(2)]
        # divide by a DEVICE tensor as the reference does (`/ normalize_factor`, :90-91): a true IEEE divide.  `gx / (w - 1.0)` with a
        # Python scalar is evaluated as gx * (1 / (w - 1.0)) by ATen's device kernels and can land 1 ulp away; the
        # factor is built once per (device, dtype, h, w) through pinned staging, so the step still has no blocking upload
        nfc = self.__dict__.setdefault("_shift_norm_cache", {})
        nkey = (str(dev), dt, h, w)
        if nkey not in nfc:
            nfc[nkey] = L.h2d(torch.tensor([w - 1.0, h - 1.0], dtype=dt), dev)
        nf = nfc[nkey]
        grid = torch.stack((gx / nf[0] * 2.0 - 1.0, gy / nf[1] * 2.0 - 1.0), -1)
        grid = grid.to(dt)
        if (self.interpolation_mode == "bilinear" and input.is_cuda and dt == torch.float32 and c % 4 == 0
                and input.is_contiguous(memory_format=torch.channels_last)
                and not (torch.is_grad_enabled() and (input.requires_grad or grid.requires_grad))):
            out = torch.empty_like(input)               # the step's case: the adjacent frame's map is detached, forward only
            gr = grid.contiguous()
            with torch.cuda.device(dev):
                L.call("dbev_grid_sample_bilinear_nhwc", L.ptr(input), L.ptr(gr), n, c, h, w, h, w, L.ptr(out), L.stream_ptr(dev))
            return out
        return F.grid_sample(input, grid, align_corners=True, mode=self.interpolation_mode)

    def bev_encoder(self, x, return_backbone_feature=False):
        feats = self.img_bev_encoder_backbone(x)
        out = self.img_bev_encoder_neck(feats)
        return (out, feats) if return_backbone_feature else out

    def extract_img_feat(self, img, img_metas=None, return_lss_feature=False, return_backbone_feature=False):
        """bevdet_distill_more.py:370-453 with the lift+splat pair replaced by the fused op."""
        inputs = img
        B, N, _, H, W = inputs[0].shape
        N = N // 2
        imgs = [t.squeeze(2) for t in torch.split(inputs[0].view(B, N, 2, 3, H, W), 1, 2)]
        rots, trans, intrins, post_rots, post_trans = inputs[1:6]
        extra = [rots.view(B, 2, N, 3, 3), trans.view(B, 2, N, 3), intrins.view(B, 2, N, 3, 3),
                 post_rots.view(B, 2, N, 3, 3), post_trans.view(B, 2, N, 3)]
        rots, trans, intrins, post_rots, post_trans = [[p.squeeze(1) for p in torch.split(t, 1, 1)] for t in extra]
        vt = self.img_view_transformer
        bev_feat_list, depth_digit_list = [], []
        for fi, (im, intrin, post_rot, post_tran) in enumerate(zip(imgs, intrins, post_rots, post_trans)):
            tran, rot = trans[0], rots[0]            # current-frame extrinsics for both frames (:389-393)
            # the reference runs the adjacent frame under autograd and then detaches its BEV feature (:440-441): no
            # gradient ever reaches this branch, so its graph (12 GB of saved activations at bs 8) is not recorded
            with torch.set_grad_enabled(torch.is_grad_enabled() and not (self.detach and fi == 1)):
                graphed = getattr(self, "adjacent_graph", None) if (fi == 1 and self.training and not torch.is_grad_enabled()) else None
                x = graphed(im) if graphed is not None else self.image_encoder(im)     # (graphed.py: the gradient-free frame as one hipGraph)
                Bx, Nx, C, fH, fW = x.shape
                img_feat, depth_digit, depth = vt.depth_feat_and_prob(x.view(Bx * Nx, C, fH, fW), rot, tran, intrin, post_rot,
                                                                      post_tran)
                # get_geometry + lift + voxel_pooling (:411-421) in three library calls, no volume, no geom tensor
                bev_feat_list.append(vt.lift_splat_cameras(rot, tran, intrin, post_rot, post_tran, depth, img_feat))
            depth_digit_list.append(depth_digit)
        # the adjacent frame's map is detached below (:440-441): with self.detach nothing of its branch is recorded -- same values,
        # no saved activations of pre_process_net, and the warp runs on the forward-only kernel
        record = torch.is_grad_enabled() and not self.detach
        adj = lambda: torch.set_grad_enabled(record)          # (constructing it already switches the mode: one per `with`)
        if self.before and self.pre_process:
            cur = self.pre_process_net(bev_feat_list[0])[0]
            with adj():
                bev_feat_list = [cur, self.pre_process_net(bev_feat_list[1])[0]]
        with adj():
            bev_feat_list[1] = self.shift_feature(bev_feat_list[1], trans, rots)
        if self.pre_process and not self.before:
            cur = self.pre_process_net(bev_feat_list[0])[0]
            with adj():
                bev_feat_list = [cur, self.pre_process_net(bev_feat_list[1])[0]]
        if self.detach:
            bev_feat_list[1] = bev_feat_list[1].detach()
        bev_feat = torch.cat(bev_feat_list, dim=1)
        outputs = []
        if return_lss_feature:
            outputs.append(bev_feat)
        if return_backbone_feature:
            x, backbone_feature = self.bev_encoder(bev_feat, True)
            outputs.append(backbone_feature)
            outputs.insert(0, x)
        else:
            outputs.insert(0, self.bev_encoder(bev_feat))
        return outputs, depth_digit_list[0]

    def get_depth_loss(self, depth_gt, depth):
        """bevdet_distill_more.py:185-204."""
        vt = self.img_view_transformer
        B, N, H, W = depth_gt.shape
        loss_weight = (~(depth_gt == 0)).reshape(B, N, 1, H, W).expand(B, N, vt.D, H, W)
        dg = (depth_gt - vt.grid_config["dbound"][0]) / vt.grid_config["dbound"][2]
        dg = torch.clip(torch.floor(dg), 0, vt.D).to(torch.long)
        # (the reference's one_hot(num_classes=D) raises for a depth >= dbound[1]; such a pixel gets an
        #  all-zero target row here)
        logit = F.one_hot(dg.reshape(-1), num_classes=vt.D + 1)[:, :vt.D]
        logit = logit.reshape(B, N, H, W, vt.D).permute(0, 1, 4, 2, 3).to(torch.float32)
        d = depth.sigmoid().view(B, N, vt.D, H, W)
        return vt.loss_depth_weight * F.binary_cross_entropy(d, logit, weight=loss_weight.float())

    # ---- distillation -------------------------------------------------------------------------
    def add_fp_as_fg(self, mode, fg_mask, heatmaps, teacher_preds, student_preds):
        """bevdet_distill.py:846-970 for equal-size maps, fp_scale_mode 'average'."""
        dp = self.distill_params
        thres = dp["output_threshold"]
        gt_thres = dp["groundtruth_threshold"] if dp["groundtruth_threshold"] is not None else thres
        gt_max = torch.cat(list(heatmaps), dim=1).max(dim=1, keepdim=True)[0]
        t_max = torch.cat([clip_sigmoid(tp[0]["heatmap"]) for tp in teacher_preds], dim=1).max(dim=1, keepdim=True)[0].detach()
        s_max = torch.cat([sp[0]["heatmap"] for sp in student_preds], dim=1).max(dim=1, keepdim=True)[0].detach()
        if mode == "teacher":
            fp = (gt_max < gt_thres) & (t_max > thres)
        elif mode == "student":
            fp = (gt_max < gt_thres) & (s_max > thres)
        elif mode == "teacher_selected_student":
            fp = (gt_max < gt_thres) & (s_max > thres) & (t_max < gt_thres)
        else:
            raise NotImplementedError(mode)
        assert fp.shape == fg_mask.shape, "multi-resolution fp masks are outside the hot-path recipe"
        fp = ((fg_mask == 0) & fp).detach().float()
        assert dp["fp_scale_mode"] == "average", "fp_scale_mode='dfs' (python BFS) is outside the hot-path recipe"
        n_fp = fp.sum(dim=(1, 2, 3))
        fp_scale = fp / n_fp.clamp(min=1).view(-1, 1, 1, 1)
        return fp, fp_scale, n_fp

    def fgd_distill_loss(self, teacher_feat, student_feat, gt_bboxes_3d, gt_labels_3d, canvas_feat, heatmaps,
                         teacher_preds, student_preds, index):
        """bevdet_distill.py:973-1324 for foreground_mask='gt', background_mask='logical_not',
        scale_mask='combine_gt', non_empty_weight=0, affinity_mode='none'."""
        dp = self.distill_params
        assert dp["foreground_mask"] == "gt" and dp["background_mask"] == "logical_not"
        assert dp["scale_mask"] == "combine_gt" and dp["non_empty_weight"] == 0
        assert dp["affinity_mode"][index] == "none" and dp["context_length"] == 0
        # the rasteriser emits the [y, x] mask layout of transpose_mask=False (every shipped recipe); bevdet_distill.py:1022-1027
        assert not dp.get("transpose_mask", False), "transpose_mask=True is outside the hot-path recipe"
        teacher_feat = self.teacher_adaptations[index](teacher_feat)
        adapt = self.channel_wise_adaptations[index]
        # 'head' recipe (1x1-conv adaptation): GEMM + loss reductions in one MFMA kernel, no adapted tensor in memory
        fused = self.fused_adapt_mse and not dp["channel_mask"] and fused_adapt_eligible(adapt, student_feat, teacher_feat)
        if (not fused and self.fused_adapt_mse and not dp["channel_mask"] and type(adapt).__name__ in ("Conv2d", "BiasSumConv2d") and isinstance(adapt, nn.Conv2d) and adapt.kernel_size == (1, 1)
                and getattr(self, "channels_last", False) and student_feat.is_cuda):
            L.note_fallback("adapt_mse", f"{adapt.in_channels}->{adapt.out_channels}, layouts {student_feat.stride()} / {teacher_feat.stride()}")
        if not fused:
            student_feat = adapt(student_feat)
            assert teacher_feat.shape == student_feat.shape
        B, C, H, W = teacher_feat.shape
        fg, fg_scale, bg_scale = self._fg_raster(H, W, [b.tensor for b in gt_bboxes_3d], student_feat.device)
        fp = fp_scale = n_fp = None
        use_fp = dp["fp_as_foreground"][index] != "none" and self._epoch >= dp["fp_epoch"]
        if use_fp:
            fp, fp_scale, n_fp = self.add_fp_as_fg(dp["fp_as_foreground"][index], fg, heatmaps, teacher_preds, student_preds)
        kw = dict(w_fg=_pick(dp["fg_feat_loss_weights"], index), w_bg=_pick(dp["bg_feat_loss_weights"], index),
                  spatial_t=dp["spatial_t"], channel_t=dp["channel_t"], s_ratio=dp["spatial_student_ratio"],
                  spatial_att=_pick(dp["spatial_attentions"], index), spatial_mask=dp["spatial_mask"],
                  channel_mask=dp["channel_mask"], fp=fp, fp_scale=fp_scale, n_fp=n_fp, w_fp=dp["fp_weight"])
        if fused:
            losses, att, _, pools = fgd_feature_losses_fused_adapt(student_feat, adapt, teacher_feat, fg, fg_scale, bg_scale, **kw)
        else:
            losses, att, _, pools = fgd_feature_losses(student_feat, teacher_feat, fg, fg_scale, bg_scale, **kw)
        if dp["spatial_mask"]:
            if pools is not None:                  # mean over channels fused into the attention / dS kernels
                t_pool, s_pool = pools
            else:
                t_pool = torch.mean(teacher_feat, [1], keepdim=True)
                s_pool = torch.mean(student_feat, [1], keepdim=True)
            losses["kd_spatial_loss"] = self.spatial_criterion(
                t_pool, self.spatial_wise_adaptations[index](s_pool)).sum() * (_pick(dp["spatial_loss_weights"], index) / B)
        return losses

    def forward_distill(self, points, img_metas, gt_bboxes_3d, gt_labels_3d, img_feats, lss_feat, bev_backbone_feats,
                        preds, heatmaps):
        """bevdet_distill.py:1411-1507."""
        dp = self.distill_params
        with torch.no_grad():
            t_neck, canvas, t_backbone = self.teacher_model.extract_pts_feat(
                points, img_feats=None, img_metas=img_metas, return_canvas=True, return_backbone_feature=True)
            if not isinstance(t_neck, (list, tuple)):
                t_neck = [t_neck]
            # the teacher's predictions are read in ONE place: the heat maps of add_fp_as_fg (bevdet_distill.py:846-870).  The
            # other five branches of each task head (30 of its 36 branch stacks) and, when no position uses the false-positive
            # term in this epoch, the whole head are skipped: unused values, identical losses.
            # DBEV_TEACHER_FULL_HEAD=1 evaluates all 36 branch stacks as the reference does (same losses; for A/B timing).
            need = any(m != "none" for m in dp["fp_as_foreground"]) and self._epoch >= dp["fp_epoch"]
            if os.environ.get("DBEV_TEACHER_FULL_HEAD") == "1":
                teacher_preds = self.teacher_model.pts_bbox_head(t_neck)
            else:
                teacher_preds = self.teacher_model.pts_bbox_head(t_neck, only=("heatmap",)) if need else None
        out = {}
        for index, (spos, tpos) in enumerate(zip(dp["student_feat_pos"], dp["teacher_feat_pos"])):
            if spos == "head":
                s_feat = img_feats[0]
            elif spos == "lss":
                s_feat = lss_feat
            elif spos.startswith("backbone"):
                if self._epoch < dp["multi_scale_epoch"]:
                    continue
                s_feat = bev_backbone_feats[int(spos[-1])]
            else:
                raise NotImplementedError(spos)
            if tpos == "head":
                t_feat = t_neck[0]
            elif tpos.startswith("backbone"):
                t_feat = t_backbone[int(tpos[-1])]
            elif tpos == "canvas":
                t_feat = canvas
            else:
                raise NotImplementedError(tpos)
            ld = self.fgd_distill_loss(t_feat, s_feat, gt_bboxes_3d, gt_labels_3d, canvas, heatmaps,
                                       teacher_preds, preds, index)
            for k, v in ld.items():
                out[f"{k}_{spos}_{tpos}"] = v
        return out

    def forward_train(self, points=None, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None, gt_labels=None,
                      gt_bboxes=None, img_inputs=None, proposals=None, gt_bboxes_ignore=None):
        """bevdet_distill_more.py:457-522 -> dict of scalar losses."""
        (img_feats, lss_feat, bev_backbone_feats), depth = self.extract_img_feat(
            img_inputs, img_metas, return_lss_feature=True, return_backbone_feature=True)
        img_feats = [img_feats]
        depth_gt = img_inputs[-1]
        B, N, H, W = depth_gt.shape
        depth_gt = depth_gt.view(B, 2, N // 2, H, W)[:, 0]
        losses = dict(loss_depth=self.get_depth_loss(depth_gt, depth))
        preds, (losses_pts, heatmaps, anno_boxes, inds, masks) = self.forward_pts_train(
            img_feats, gt_bboxes_3d, gt_labels_3d, img_metas, gt_bboxes_ignore, get_preds=True, get_targets=True)
        losses.update(losses_pts)
        losses.update(self.forward_distill(points, img_metas, gt_bboxes_3d, gt_labels_3d, img_feats, lss_feat,
                                           bev_backbone_feats, preds, heatmaps))
        return losses


# ---- the other detectors of the reference's config surface --------------------------------------
# One student/distillation implementation above; the siblings differ in the number of frames and in whether the
# view transformer predicts (and is supervised on) depth.

@MODELS.register_module()
class BEVDepth4D(BEVDepth4DDistill):
    """mmdet3d/models/detectors/bevdet.py:509-680: the two-frame BEVDepth student on its own -- no teacher, no
    distillation; forward_train = depth loss + CenterHead losses (:631-680)."""

    def __init__(self, **kwargs):
        for k in ("teacher_config", "teacher_ckpt", "distill_type", "distill_params"):
            assert kwargs.get(k) is None, f"BEVDepth4D takes no {k}"
        super().__init__(**kwargs)

    def forward_train(self, points=None, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None, gt_labels=None,
                      gt_bboxes=None, img_inputs=None, proposals=None, gt_bboxes_ignore=None):
        (img_feats,), depth = self.extract_img_feat(img_inputs, img_metas)
        depth_gt = img_inputs[-1]
        B, N, H, W = depth_gt.shape
        depth_gt = depth_gt.view(B, 2, N // 2, H, W)[:, 0]
        losses = dict(loss_depth=self.get_depth_loss(depth_gt, depth))
        losses.update(self.forward_pts_train([img_feats], gt_bboxes_3d, gt_labels_3d, img_metas, gt_bboxes_ignore))
        return losses


@MODELS.register_module()
class BEVDepthDistill(BEVDepth4DDistill):
    """bevdet_distill_more.py:168-330: single-frame BEVDepth student (ViewTransformerLSSBEVDepth: BEV feature + depth
    logits) distilled from the LiDAR teacher.  img_inputs = (imgs[B,N,3,H,W], rots, trans, intrins, post_rots,
    post_trans, depth_gt[B,N,h,w])."""

    def extract_img_feat(self, img, img_metas=None, return_lss_feature=False, return_backbone_feature=False):
        x = self.image_encoder(img[0])                                   # [B, N, C, fH, fW]
        bev_feat, depth = self.img_view_transformer([x] + list(img[1:]))  # :245
        outputs = []
        if return_lss_feature:
            outputs.append(bev_feat)
        if return_backbone_feature:
            x, backbone_feature = self.bev_encoder(bev_feat, True)
            outputs.append(backbone_feature)
            outputs.insert(0, x)
        else:
            outputs.insert(0, self.bev_encoder(bev_feat))
        return outputs, depth

    def forward_train(self, points=None, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None, gt_labels=None,
                      gt_bboxes=None, img_inputs=None, proposals=None, gt_bboxes_ignore=None):
        (img_feats, lss_feat, bev_backbone_feats), depth = self.extract_img_feat(
            img_inputs, img_metas, return_lss_feature=True, return_backbone_feature=True)
        img_feats = [img_feats]
        losses = dict(loss_depth=self.get_depth_loss(img_inputs[-1], depth))            # :300-302
        preds, (losses_pts, heatmaps, anno_boxes, inds, masks) = self.forward_pts_train(
            img_feats, gt_bboxes_3d, gt_labels_3d, img_metas, gt_bboxes_ignore, get_preds=True, get_targets=True)
        losses.update(losses_pts)
        losses.update(self.forward_distill(points, img_metas, gt_bboxes_3d, gt_labels_3d, img_feats, lss_feat,
                                           bev_backbone_feats, preds, heatmaps))
        return losses


@MODELS.register_module()
class BEVDetDistill(BEVDepthDistill):
    """bevdet_distill.py:155-1560: single-frame BEVDet student (ViewTransformerLiftSplatShoot: no depth head, no depth
    supervision) distilled from the LiDAR teacher."""

    def extract_img_feat(self, img, img_metas=None, return_lss_feature=False, return_backbone_feature=False):
        x = self.image_encoder(img[0])
        bev_feat = self.img_view_transformer([x] + list(img[1:6]))       # bevdet.py:58-60
        outputs = []
        if return_lss_feature:
            outputs.append(bev_feat)
        feats = self.img_bev_encoder_backbone(bev_feat)
        if return_backbone_feature:
            outputs.append(feats)
        outputs.insert(0, self.img_bev_encoder_neck(feats))
        return outputs

    def forward_train(self, points=None, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None, gt_labels=None,
                      gt_bboxes=None, img_inputs=None, proposals=None, gt_bboxes_ignore=None):
        img_feats, lss_feat, bev_backbone_feats = self.extract_img_feat(
            img_inputs, img_metas, return_lss_feature=True, return_backbone_feature=True)
        img_feats = [img_feats]
        preds, (losses_pts, heatmaps, anno_boxes, inds, masks) = self.forward_pts_train(
            img_feats, gt_bboxes_3d, gt_labels_3d, img_metas, gt_bboxes_ignore, get_preds=True, get_targets=True)
        losses = dict(losses_pts)
        losses.update(self.forward_distill(points, img_metas, gt_bboxes_3d, gt_labels_3d, img_feats, lss_feat,
                                           bev_backbone_feats, preds, heatmaps))
        return losses


@MODELS.register_module()
class BEVDet4DDistill(BEVDepth4DDistill):
    """bevdet_distill_more.py:15-166: two-frame BEVDet student (plain LSS view transformer: one 1x1 depthnet conv gives
    the D depth logits and the C context channels, :121-126; no depth supervision) distilled from the LiDAR teacher."""

    def extract_img_feat(self, img, img_metas=None, return_lss_feature=False, return_backbone_feature=False):
        inputs = img
        B, N, _, H, W = inputs[0].shape
        N = N // 2
        imgs = [t.squeeze(2) for t in torch.split(inputs[0].view(B, N, 2, 3, H, W), 1, 2)]
        rots, trans, intrins, post_rots, post_trans = inputs[1:6]
        extra = [rots.view(B, 2, N, 3, 3), trans.view(B, 2, N, 3), intrins.view(B, 2, N, 3, 3),
                 post_rots.view(B, 2, N, 3, 3), post_trans.view(B, 2, N, 3)]
        rots, trans, intrins, post_rots, post_trans = [[p.squeeze(1) for p in torch.split(t, 1, 1)] for t in extra]
        vt = self.img_view_transformer
        bev_feat_list = []
        for fi, (im, intrin, post_rot, post_tran) in enumerate(zip(imgs, intrins, post_rots, post_trans)):
            tran, rot = trans[0], rots[0]
            with torch.set_grad_enabled(torch.is_grad_enabled() and not (self.detach and fi == 1)):
                graphed = getattr(self, "adjacent_graph", None) if (fi == 1 and self.training and not torch.is_grad_enabled()) else None
                x = graphed(im) if graphed is not None else self.image_encoder(im)     # (graphed.py: the gradient-free frame as one hipGraph)
                Bx, Nx, C, fH, fW = x.shape
                x = vt.depthnet(x.view(Bx * Nx, C, fH, fW))
                depth = vt.get_depth_dist(x[:, :vt.D])
                img_feat = x[:, vt.D:(vt.D + vt.numC_Trans)]
                bev_feat_list.append(vt.lift_splat_cameras(rot, tran, intrin, post_rot, post_tran, depth, img_feat))
        # the adjacent frame's map is detached below (:440-441): with self.detach nothing of its branch is recorded -- same values,
        # no saved activations of pre_process_net, and the warp runs on the forward-only kernel
        record = torch.is_grad_enabled() and not self.detach
        adj = lambda: torch.set_grad_enabled(record)          # (constructing it already switches the mode: one per `with`)
        if self.before and self.pre_process:
            cur = self.pre_process_net(bev_feat_list[0])[0]
            with adj():
                bev_feat_list = [cur, self.pre_process_net(bev_feat_list[1])[0]]
        with adj():
            bev_feat_list[1] = self.shift_feature(bev_feat_list[1], trans, rots)
        if self.pre_process and not self.before:
            cur = self.pre_process_net(bev_feat_list[0])[0]
            with adj():
                bev_feat_list = [cur, self.pre_process_net(bev_feat_list[1])[0]]
        if self.detach:
            bev_feat_list[1] = bev_feat_list[1].detach()
        bev_feat = torch.cat(bev_feat_list, dim=1)
        outputs = []
        if return_lss_feature:
            outputs.append(bev_feat)
        feats = self.img_bev_encoder_backbone(bev_feat)
        if return_backbone_feature:
            outputs.append(feats)
        outputs.insert(0, self.img_bev_encoder_neck(feats))
        return outputs

    forward_train = BEVDetDistill.forward_train          # no depth term (inherits BEVDetDistill.forward_train, :15-16)
