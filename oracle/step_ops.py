"""Oracle (CPU, torch) restatements of the step functions that sit between the hot kernels
-- TEST INFRASTRUCTURE ONLY, never imported by the product package.

  shift_feature       mmdet3d/models/detectors/bevdet_distill_more.py:41-94
  get_depth_loss      mmdet3d/models/detectors/bevdet_distill_more.py:185-204
  add_fp_as_fg        mmdet3d/models/detectors/bevdet_distill.py:846-920 (equal-size maps, 'average' scale)
  centerhead_loss     mmdet3d/models/dense_heads/centerpoint_head.py:615-686 with the mmdet==2.24.0 losses it is
                      configured with (GaussianFocalLoss, L1Loss: un-vendored third party, published definitions)
  get_geometry        mmdet3d/models/necks/view_transformer_mine.py:111-139 (torch, the reference's own matmul order)
  pillar_feature_net  mmdet3d/models/voxel_encoders/pillar_encoder.py:283-338 (single PFN layer)

Each one keeps the reference's operation ORDER (torch.inverse + broadcast matmul, in-place clip_sigmoid, python loops
over samples) so that it is an independent check of the product's re-ordered / fused device versions.

Pinning: every function here is compared on the CPU with outputs of the reference's own files imported by path
(tests/golden/make_golden.py sections shift_depth / fgd / centerloss / pfn -> tests/test_oracle_step_ops.py).
"""
import torch
import torch.nn.functional as F


# ---- BEVDet4D temporal alignment ---------------------------------------------------------------
def _homogeneous(rot, tran):
    n, v = tran.shape[:2]
    m = torch.zeros((n, v, 4, 4), dtype=rot.dtype, device=rot.device)
    m[..., :3, :3] = rot
    m[..., :3, 3] = tran
    m[..., 3, 3] = 1
    return m


def shift_feature(x, trans, rots, dx, bx, mode="bilinear"):
    """x f32[n,c,h,w]; trans/rots = [current, adjacent] camera->lidar translations [n,v,3] / rotations [n,v,3,3];
    dx, bx = view-transformer cell size / first cell centre.  -> grid_sample'd feature, align_corners=True."""
    n, c, h, w = x.shape
    col = torch.linspace(0, w - 1, w, dtype=x.dtype, device=x.device).view(1, w).expand(h, w)
    row = torch.linspace(0, h - 1, h, dtype=x.dtype, device=x.device).view(h, 1).expand(h, w)
    pix = torch.stack((col, row, torch.ones_like(col)), -1).view(1, h, w, 3, 1).expand(n, h, w, 3, 1)
    cur = _homogeneous(rots[0], trans[0])
    adj = _homogeneous(rots[1], trans[1])
    l02l1 = cur.matmul(torch.inverse(adj))[:, 0].view(n, 1, 1, 4, 4)          # :66, camera 0 only
    keep = torch.tensor([0, 1, 3], device=x.device)
    l02l1 = l02l1.index_select(3, keep).index_select(4, keep)                  # :76 drop z
    f2b = torch.zeros((3, 3), dtype=x.dtype, device=x.device)
    f2b[0, 0], f2b[1, 1] = dx[0], dx[1]
    f2b[0, 2] = bx[0] - dx[0] / 2.0
    f2b[1, 2] = bx[1] - dx[1] / 2.0
    f2b[2, 2] = 1
    f2b = f2b.view(1, 3, 3)
    tf = torch.inverse(f2b).matmul(l02l1).matmul(f2b)                          # :86
    moved = tf.matmul(pix)                                                     # :89 broadcast matmul
    norm = torch.tensor([w - 1.0, h - 1.0], dtype=x.dtype, device=x.device)
    grid = moved[..., :2, 0] / norm.view(1, 1, 1, 2) * 2.0 - 1.0
    return F.grid_sample(x, grid.to(x.dtype), align_corners=True, mode=mode)


# ---- depth supervision ---------------------------------------------------------------------------
def get_depth_loss(depth_gt, depth_logits, dbound, D, loss_depth_weight):
    """depth_gt f32[B,N,H,W] (0 = no lidar return), depth_logits f32[B*N,D,H,W] -> scalar.
    (one_hot(num_classes=D) as in the reference: a gt depth >= dbound[1] raises, as it does there.)"""
    B, N, H, W = depth_gt.shape
    weight = (~(depth_gt == 0)).reshape(B, N, 1, H, W).expand(B, N, D, H, W)
    bins = torch.clip(torch.floor((depth_gt - dbound[0]) / dbound[2]), 0, D).to(torch.long)
    target = F.one_hot(bins.reshape(-1), num_classes=D).reshape(B, N, H, W, D).permute(0, 1, 4, 2, 3).to(torch.float32)
    prob = depth_logits.sigmoid().view(B, N, D, H, W)
    return loss_depth_weight * F.binary_cross_entropy(prob, target, weight=weight)


# ---- false-positive mask ------------------------------------------------------------------------
def clip_sigmoid_(x, eps=1e-4):
    return torch.clamp(x.sigmoid_(), min=eps, max=1 - eps)          # models/utils/clip_sigmoid.py (in place)


def add_fp_as_fg(mode, fg_mask, gt_heatmaps, teacher_heatmap_logits, student_heatmaps, thres, gt_thres=None):
    """Lists (one entry per task) of [B, cls, H, W] maps.  The teacher logits are sigmoided IN PLACE (as the reference
    does through clip_sigmoid); the student maps are taken as they are (already clipped sigmoids, :867-869).
    -> (fp_mask f32[B,1,H,W], fp_scale_mask, n_fp f32[B])"""
    gt_thres = thres if gt_thres is None else gt_thres
    gt_max = torch.cat(list(gt_heatmaps), dim=1).max(dim=1, keepdim=True)[0]
    t_max = torch.cat([clip_sigmoid_(t) for t in teacher_heatmap_logits], dim=1).max(dim=1, keepdim=True)[0].detach()
    s_max = torch.cat(list(student_heatmaps), dim=1).max(dim=1, keepdim=True)[0].detach()
    assert gt_max.shape == t_max.shape == s_max.shape == fg_mask.shape, "equal-size maps only"
    if mode == "teacher":
        fp = torch.logical_and(gt_max < gt_thres, t_max > thres)
    elif mode == "student":
        fp = torch.logical_and(gt_max < gt_thres, s_max > thres)
    elif mode == "teacher_selected_student":
        fp = torch.logical_and(torch.logical_and(gt_max < gt_thres, s_max > thres), t_max < gt_thres)
    elif mode == "teacher+teacher_selected_student":
        a = torch.logical_and(gt_max < gt_thres, t_max > thres)
        b = torch.logical_and(torch.logical_and(gt_max < gt_thres, s_max > thres), t_max < gt_thres)
        fp = torch.logical_or(a, b)
    else:
        raise NotImplementedError(mode)
    fp = torch.logical_and(fg_mask == 0, fp).detach().float()
    scale = torch.zeros_like(fp)
    for b in range(fp.shape[0]):                                       # :914-916 ('average')
        scale[b][fp[b] > 0] = 1.0 / torch.sum(fp[b])
    return fp, scale, torch.sum(fp, dim=(1, 2, 3))


# ---- CenterHead loss --------------------------------------------------------------------------------
def _mean_with_avg_factor(loss, weight, avg_factor):
    """mmdet weight_reduce_loss, reduction='mean' with avg_factor (mmdet/models/losses/utils.py, 2.24.0: the divisor
    carries +float32 eps)."""
    if weight is not None:
        loss = loss * weight
    return loss.sum() / (avg_factor + torch.finfo(torch.float32).eps)


def gaussian_focal_loss(pred, target, alpha=2.0, gamma=4.0):
    eps = 1e-12
    pos = -(pred + eps).log() * (1 - pred).pow(alpha) * target.eq(1)
    neg = -(1 - pred + eps).log() * pred.pow(alpha) * (1 - target).pow(gamma)
    return pos + neg


def centerhead_loss(preds_dicts, heatmaps, anno_boxes, inds, masks, code_weights, loss_bbox_weight=0.25,
                    loss_cls_weight=1.0, loss_prefix=""):
    """preds_dicts: per task [dict(reg, height, dim, rot, vel, heatmap)] of [B, c, H, W] tensors (heatmap = logits; it is
    replaced by its clipped sigmoid IN PLACE like the reference); targets per task as CenterHead.get_targets returns
    them.  task_specific=True, GaussianFocalLoss(mean), L1Loss(mean, loss_weight) -> dict of 36 scalars."""
    out = {}
    names, edges = ["xy", "z", "whl", "yaw", "vel"], [0, 2, 3, 6, 8, 10]
    for t, pd in enumerate(preds_dicts):
        p = pd[0]
        p["heatmap"] = clip_sigmoid_(p["heatmap"])
        num_pos = heatmaps[t].eq(1).float().sum().item()
        out[f"{loss_prefix}task{t}.loss_heatmap"] = loss_cls_weight * _mean_with_avg_factor(
            gaussian_focal_loss(p["heatmap"], heatmaps[t]), None, max(num_pos, 1))
        box = torch.cat((p["reg"], p["height"], p["dim"], p["rot"], p["vel"]), dim=1)
        box = box.permute(0, 2, 3, 1).contiguous()
        box = box.view(box.size(0), -1, box.size(3))
        ind = inds[t]
        box = box.gather(1, ind.unsqueeze(2).expand(ind.size(0), ind.size(1), box.size(2)))
        target = anno_boxes[t]
        num = masks[t].float().sum()
        m = masks[t].unsqueeze(2).expand_as(target).float()
        m = m * (~torch.isnan(target)).float()
        w = m * m.new_tensor(code_weights)
        for r, nm in enumerate(names):
            sl = slice(edges[r], edges[r + 1])
            out[f"{loss_prefix}task{t}.loss_{nm}"] = loss_bbox_weight * _mean_with_avg_factor(
                torch.abs(box[..., sl] - target[..., sl]), w[..., sl], num + 1e-4)
    return out


# ---- LSS geometry in the reference's own op order -----------------------------------------------------
def get_geometry(frustum, rots, trans, intrins, post_rots, post_trans):
    """view_transformer_mine.py:111-139 with torch CPU ops exactly as written there (torch.inverse, broadcast matmul)."""
    B, N, _ = trans.shape
    points = frustum - post_trans.view(B, N, 1, 1, 1, 3)
    points = torch.inverse(post_rots).view(B, N, 1, 1, 1, 3, 3).matmul(points.unsqueeze(-1))
    points = torch.cat((points[:, :, :, :, :, :2] * points[:, :, :, :, :, 2:3], points[:, :, :, :, :, 2:3]), 5)
    combine = rots.matmul(torch.inverse(intrins))
    points = combine.view(B, N, 1, 1, 1, 3, 3).matmul(points).squeeze(-1)
    return points + trans.view(B, N, 1, 1, 1, 3)


# ---- DynamicPillarFeatureNet (one PFN layer) on the C oracle's scatter ----------------------------------
def pillar_feature_net(points, coors4, weight, bn_weight, bn_bias, bn_mean, bn_var, eps, voxel_size, pc_range,
                       training):
    """points f32[N,F], coors4 int32[N,4]=(b,z,y,x) (all rows valid) -> (voxel_feats [M,C], voxel_coors int32[M,4]).
    Per sample: cluster mean by sorted-unique scatter (oracle/voxel.c), decoration [F + 3 + 2], Linear(no bias) ->
    BatchNorm1d (batch statistics over ALL points of the batch when training) -> ReLU -> scatter max."""
    import numpy as np
    from oracle import voxel as OV
    pts = points.detach().cpu().numpy().astype(np.float32)
    co = coors4.detach().cpu().numpy().astype(np.int32)
    B = int(co[-1, 0]) + 1
    vx, vy = float(voxel_size[0]), float(voxel_size[1])
    x_off, y_off = vx / 2 + float(pc_range[0]), vy / 2 + float(pc_range[1])
    deco, groups = [], []
    for b in range(B):
        sel = np.nonzero(co[:, 0] == b)[0]
        p, c3 = pts[sel], np.ascontiguousarray(co[sel, 1:])
        mean, _, cmap, _ = OV.dynamic_scatter_forward(p, c3, "mean")
        f = torch.from_numpy(p)
        f_cluster = f[:, :3] - torch.from_numpy(mean[cmap][:, :3])
        cz = torch.from_numpy(c3.astype(np.float32))
        f_center = torch.stack([f[:, 0] - (cz[:, 2] * vx + x_off), f[:, 1] - (cz[:, 1] * vy + y_off)], 1)
        deco.append(torch.cat([f, f_cluster, f_center], -1))
        groups.append((sel, c3))
    x = torch.cat(deco, 0)                       # samples are contiguous in the reference's batch too
    y = F.linear(x, weight)
    if training:
        y = F.batch_norm(y, None, None, bn_weight, bn_bias, True, 0.0, eps)
    else:
        y = F.batch_norm(y, bn_mean, bn_var, bn_weight, bn_bias, False, 0.0, eps)
    y = F.relu(y).detach().numpy()
    vf, vc, off = [], [], 0
    for b, (sel, c3) in enumerate(groups):
        red, oc, _, _ = OV.dynamic_scatter_forward(y[off:off + len(sel)], c3, "max")
        off += len(sel)
        vf.append(red)
        vc.append(np.concatenate([np.full((oc.shape[0], 1), b, np.int32), oc], 1))
    return np.concatenate(vf, 0), np.concatenate(vc, 0)
