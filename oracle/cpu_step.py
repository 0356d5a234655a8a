"""CPU restatement of the whole distillation training step -- TEST / BASELINE INFRASTRUCTURE.

``CpuBEVDepth4DDistill`` is the reference's step executed the reference's way on the host:
the dense torch modules are the product's own (plain torch, run on CPU), and every hot op the
product sends to the HIP library is replaced by the reference's CPU op sequence:

  lift + splat      materialised volume + argsort/cumsum voxel_pooling   (oracle/lss_torch.py,
                    bevdet_distill_more.py:413-421, view_transformer_mine.py:141-181)
  teacher pillars   per-sample dynamic voxelize + sorted-unique scatter (mean, max) + canvas
                    scatter                                              (oracle/voxel.c)
  fg masks          numpy points_in_rbbox rasterisation                  (oracle/distill.py)
  FGD losses        (S-T)^2 materialised and reduced once per loss term, exactly the unfused
                    expression of bevdet_distill.py:1253-1262,1282-1287

Used by (a) bench.py's cpu_baseline leg ("port", everything on the CPU), (b) the end-to-end
parity test that compares every loss value of the HIP path against this path on identical
weights and inputs.  The op replacements are device agnostic: with the model on a GPU the
dense modules run there (same MIOpen kernels as the product, so only the hot ops differ) while
the numpy/C oracle ops still run on the host.  The product package never imports this module.
"""
import numpy as np
import torch
import torch.nn.functional as F

from distill_bev_amd import detectors as D
from distill_bev_amd.center_head import clip_sigmoid
from oracle import center_targets as OCT
from oracle import dcn as ODCN
from oracle import distill as OD
from oracle import lss_torch as OT
from oracle import step_ops as OS
from oracle import voxel as OV


class CpuDynamicCenterPoint(D.DynamicCenterPoint):
    """dynamic_centerpoint.py:43-93 + pillar_encoder.py:283-338 with the reference's op sequence."""

    @torch.no_grad()
    def extract_pts_feat(self, pts, img_feats=None, img_metas=None, return_canvas=False, return_backbone_feature=False):
        vl = self.pts_voxel_layer
        enc = self.pts_voxel_encoder
        dev = pts[0].device          # host ops below; tensors return to `dev` for the dense modules
        feats_all, coors_all = [], []
        for b, p in enumerate(pts):
            pn = p.detach().cpu().numpy().astype(np.float32)
            co = OV.dynamic_voxelize(pn, vl.voxel_size, vl.point_cloud_range)          # (z, y, x) / -1
            mean, _, cmap, _ = OV.dynamic_scatter_forward(pn, co, "mean")               # cluster_scatter
            valid = cmap >= 0
            pmean = np.zeros_like(pn)
            pmean[valid] = mean[cmap[valid]]
            f = torch.from_numpy(pn)
            f_cluster = f[:, :3] - torch.from_numpy(pmean[:, :3])
            cz = torch.from_numpy(co.astype(np.float32))
            f_center = torch.stack([f[:, 0] - (cz[:, 2] * enc.vx + enc.x_offset),
                                    f[:, 1] - (cz[:, 1] * enc.vy + enc.y_offset)], 1)
            x = torch.cat([f, f_cluster, f_center], dim=-1)
            feats_all.append(x)
            coors_all.append(co)
        x = torch.cat(feats_all, 0)
        point_feats = enc.pfn_layers[0](x.to(dev)).cpu()                                 # Linear+BN1d+ReLU
        off = 0
        vf, vc = [], []
        for b, co in enumerate(coors_all):
            n = co.shape[0]
            red, oc, _, _ = OV.dynamic_scatter_forward(point_feats[off:off + n].numpy(), co, "max")   # pfn_scatter
            vf.append(torch.from_numpy(red))
            vc.append(np.concatenate([np.full((oc.shape[0], 1), b, np.int32), oc], 1))
            off += n
        voxel_features = torch.cat(vf, 0)
        coors = np.concatenate(vc, 0)
        canvas = torch.from_numpy(OV.pillars_scatter(voxel_features.numpy(), coors, len(pts),
                                                     self.pts_middle_encoder.ny, self.pts_middle_encoder.nx)).to(dev)
        return self._backbone_neck(canvas, [], return_canvas, return_backbone_feature)


class CpuBEVDepth4DDistill(D.BEVDepth4DDistill):
    def _cpu_teacher(self):
        t = self.teacher_model
        if not isinstance(t, CpuDynamicCenterPoint):
            t.__class__ = CpuDynamicCenterPoint
        return t

    def extract_img_feat(self, img, img_metas=None, return_lss_feature=False, return_backbone_feature=False):
        vt = self.img_view_transformer
        # the reference sequence: materialise the volume, then argsort + cumsum voxel_pooling
        def ref_lift_splat(rot, tran, intrin, post_rot, post_tran, depth, feat):
            # the reference's own geometry op order (torch.inverse + broadcast matmul), not the product's multiply-adds
            geom = OS.get_geometry(vt.frustum, rot, tran, intrin, post_rot, post_tran)
            return OT.voxel_pooling_cumsum(geom, OT.lift(depth, feat, geom.shape[0], geom.shape[1]), vt.dx, vt.bx, vt.nx)
        vt.lift_splat_cameras = ref_lift_splat
        return super().extract_img_feat(img, img_metas, return_lss_feature, return_backbone_feature)

    # the remaining step functions run their oracle restatements (oracle/step_ops.py, pinned against the imported
    # reference files) -- nothing below is inherited from the product class
    def shift_feature(self, input, trans, rots):
        vt = self.img_view_transformer
        return OS.shift_feature(input, trans, rots, vt.dx, vt.bx, self.interpolation_mode)

    def get_depth_loss(self, depth_gt, depth):
        vt = self.img_view_transformer
        return OS.get_depth_loss(depth_gt, depth, vt.grid_config["dbound"], vt.D, vt.loss_depth_weight)

    def add_fp_as_fg(self, mode, fg_mask, heatmaps, teacher_preds, student_preds):
        dp = self.distill_params
        return OS.add_fp_as_fg(mode, fg_mask, heatmaps, [tp[0]["heatmap"] for tp in teacher_preds],
                               [sp[0]["heatmap"] for sp in student_preds], dp["output_threshold"],
                               dp["groundtruth_threshold"])

    def fgd_distill_loss(self, teacher_feat, student_feat, gt_bboxes_3d, gt_labels_3d, canvas_feat, heatmaps,
                         teacher_preds, student_preds, index):
        dp = self.distill_params
        teacher_feat = self.teacher_adaptations[index](teacher_feat)
        student_feat = self.channel_wise_adaptations[index](student_feat)
        B, C, H, W = student_feat.shape
        tc = self.pts_bbox_head.train_cfg
        fg, fgs, bgs = OD.foreground_scale_mask(H, W, [b.tensor.numpy() for b in gt_bboxes_3d], tc["grid_size"],
                                                tc["point_cloud_range"], tc["voxel_size"])
        dev = student_feat.device
        fg, fg_scale, bg_scale = torch.from_numpy(fg).to(dev), torch.from_numpy(fgs).to(dev), torch.from_numpy(bgs).to(dev)
        S_T, C_T, s_ratio = dp["spatial_t"], dp["channel_t"], dp["spatial_student_ratio"]
        t_att = torch.softmax(torch.mean(torch.abs(teacher_feat), [1]).view(B, -1) / S_T, dim=1) * H * W
        s_att = torch.softmax(torch.mean(torch.abs(student_feat), [1]).view(B, -1) / S_T, dim=1) * H * W
        c_att = (torch.softmax(torch.mean(torch.abs(teacher_feat), [2, 3]) / C_T, dim=1) * C).view(B, C, 1, 1).detach()
        sa = D._pick(dp["spatial_attentions"], index)
        att = t_att if sa == "teacher" else (t_att + s_att * s_ratio) / (1 + s_ratio)
        att = att.view(B, 1, H, W).detach()
        bg = fg.logical_not()
        use_fp = dp["fp_as_foreground"][index] != "none" and self._epoch >= dp["fp_epoch"]
        if use_fp:
            fp, fp_scale, n_fp = self.add_fp_as_fg(dp["fp_as_foreground"][index], fg, heatmaps, teacher_preds, student_preds)
            bg[fp != 0] = 0
            n_bg = H * W - torch.sum(fg, dim=(1, 2, 3))
            bg_scale = bg_scale.clone()
            for b in range(B):
                bg_scale[b][:] = 1.0 / (n_bg[b] - n_fp[b]) if n_bg[b] > n_fp[b] else 0
        scale = torch.maximum(fg_scale, bg_scale)
        fg_m = fg * scale
        bg_m = bg * scale
        if dp["spatial_mask"]:
            fg_m = fg_m * att
            bg_m = bg_m * att
        w_fg, w_bg = D._pick(dp["fg_feat_loss_weights"], index), D._pick(dp["bg_feat_loss_weights"], index)
        out = {"kd_fg_feat_loss": (self.feat_criterion(student_feat, teacher_feat) * fg_m).sum() * w_fg / B,
               "kd_bg_feat_loss": (self.feat_criterion(student_feat, teacher_feat) * bg_m).sum() * w_bg / B}
        if dp["spatial_mask"]:
            t_pool = torch.mean(teacher_feat, [1], keepdim=True)
            s_pool = torch.mean(student_feat, [1], keepdim=True)
            out["kd_spatial_loss"] = self.spatial_criterion(
                t_pool, self.spatial_wise_adaptations[index](s_pool)).sum() * D._pick(dp["spatial_loss_weights"], index) / B
        if use_fp:
            fp_m = fp * fp_scale * att * c_att
            out["kd_fp_bg_feat_loss"] = (self.feat_criterion(student_feat, teacher_feat) * fp_m).sum() * dp["fp_weight"] / B
        return out

    def forward_distill(self, *a, **k):
        self._cpu_teacher()
        return super().forward_distill(*a, **k)


def _reference_head_loss(self, gt_bboxes_3d, gt_labels_3d, preds_dicts, get_targets=False, **kwargs):
    """CenterHead.loss (centerpoint_head.py:615-686) = host targets + the op-by-op loss of oracle/step_ops.py"""
    device = preds_dicts[0][0]["heatmap"].device
    heatmaps, anno_boxes, inds, masks = self.get_targets(gt_bboxes_3d, gt_labels_3d, device)
    loss_dict = OS.centerhead_loss(preds_dicts, heatmaps, anno_boxes, inds, masks, self.train_cfg["code_weights"],
                                   self.loss_bbox.loss_weight, self.loss_cls.loss_weight, self.loss_prefix)
    return (loss_dict, heatmaps, anno_boxes, inds, masks) if get_targets else loss_dict


def to_cpu_reference(model):
    """Re-class a built (CPU) BEVDepth4DDistill into the reference-sequence CPU variant, in place."""
    model.__class__ = CpuBEVDepth4DDistill
    for seq in model.channel_wise_adaptations:         # the reference's own nn.Upsample
        if isinstance(seq, torch.nn.Sequential) and type(seq[0]).__name__ == "UpsampleBilinearAC":
            seq[0] = torch.nn.Upsample(scale_factor=seq[0].scale_factor, mode="bilinear", align_corners=True)
    # host (numpy) target assignment of the reference instead of dbev_centerhead_targets -- for the student's head
    # and the teacher's (the distillation recipe reads the teacher head's targets too)
    for head in [getattr(model, "pts_bbox_head", None), getattr(getattr(model, "teacher_model", None), "pts_bbox_head", None)]:
        if head is not None:
            head.get_targets = OCT.get_targets.__get__(head)
            head.loss = _reference_head_loss.__get__(head)
    for m in model.modules():                          # torch restatement of DCNv2 instead of the HIP kernels
        if type(m).__name__ == "ModulatedDeformConv2dPack":
            m.__class__ = type("TorchModulatedDeformConv2dPack", (m.__class__,), {"forward": ODCN.pack_forward})
    model._cpu_teacher()
    return model
