"""TEST INFRASTRUCTURE -- host (numpy) restatement of the CenterHead target assignment:
``CenterHead.get_targets / get_targets_single`` (mmdet3d/models/dense_heads/centerpoint_head.py:366-413,447-611) and
``gaussian_radius / gaussian_2d / draw_heatmap_gaussian`` (mmdet3d/core/utils/gaussian.py:6-88).

Pinned against the reference's own ``core/utils/gaussian.py`` imported in the build container
(tests/test_config_and_model.py::test_center_head_targets_match_reference_gaussian_utils).  The product
(distill_bev_amd/center_head.py) runs ``dbev_centerhead_targets`` on the GPU and is tested against this module;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
"""
import numpy as np
import torch


def gaussian_radius(det_size, min_overlap=0.5):
    """core/utils/gaussian.py:58-88 in float32 (the reference evaluates it on 0-dim float32 tensors)."""
    f = np.float32
    height, width = f(det_size[0]), f(det_size[1])
    mo = f(min_overlap)
    b1 = height + width
    c1 = width * height * (f(1) - mo) / (f(1) + mo)
    r1 = (b1 + np.sqrt(b1 * b1 - f(4) * c1)) / f(2)
    b2 = f(2) * (height + width)
    c2 = (f(1) - mo) * width * height
    r2 = (b2 + np.sqrt(b2 * b2 - f(16) * c2)) / f(2)
    a3 = f(4) * mo
    b3 = f(-2) * mo * (height + width)
    c3 = (mo - f(1)) * width * height
    r3 = (b3 + np.sqrt(b3 * b3 - f(4) * a3 * c3)) / f(2)
    return min(r1, r2, r3)


def gaussian_2d(shape, sigma=1.0):
    """gaussian.py:6-22."""
    m, n = [(ss - 1.0) / 2.0 for ss in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_heatmap_gaussian(heatmap, center, radius, k=1):
    """gaussian.py:25-55 on a numpy heatmap [H, W] (in place)."""
    diameter = 2 * radius + 1
    g = gaussian_2d((diameter, diameter), sigma=diameter / 6)
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape[:2]
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    mh = heatmap[y - top:y + bottom, x - left:x + right]
    mg = g[radius - top:radius + bottom, radius - left:radius + right].astype(np.float32)
    if min(mg.shape) > 0 and min(mh.shape) > 0:
        np.maximum(mh, mg * k, out=mh)
    return heatmap


def get_targets_single_np(head, boxes9, labels):
    """centerpoint_head.py:447-611 for one sample.  boxes9 f32[M, 9] with GRAVITY centre
    (x, y, z_c, w, l, h, yaw, vx, vy); labels int[M]."""
    cfg = head.train_cfg
    f = np.float32
    max_objs = cfg["max_objs"] * cfg["dense_reg"]
    osf = cfg["out_size_factor"]
    grid = np.asarray(cfg["grid_size"])
    pc = np.asarray(cfg["point_cloud_range"], dtype=f)
    vs = np.asarray(cfg["voxel_size"], dtype=f)
    fm = grid[:2] // osf            # (W, H)
    heatmaps, anno_boxes, inds, masks = [], [], [], []
    flag = 0
    for names in head.class_names:
        ncls = len(names)
        sel = [np.flatnonzero(labels == (j + flag)) for j in range(ncls)]
        order = np.concatenate(sel) if sel else np.zeros((0,), np.int64)
        tb = boxes9[order]
        tc = (labels[order] + 1 - flag).astype(np.int64)
        flag += ncls
        hm = np.zeros((ncls, int(fm[1]), int(fm[0])), dtype=f)
        ab = np.zeros((max_objs, 10), dtype=f)
        ind = np.zeros((max_objs,), dtype=np.int64)
        mk = np.zeros((max_objs,), dtype=np.uint8)
        for k in range(min(tb.shape[0], max_objs)):
            cls_id = int(tc[k]) - 1
            width = f(tb[k, 3] / vs[0] / f(osf))
            length = f(tb[k, 4] / vs[1] / f(osf))
            if not (width > 0 and length > 0):
                continue
            radius = gaussian_radius((length, width), min_overlap=cfg["gaussian_overlap"])
            radius = max(cfg["min_radius"], int(radius))
            cx = f(f(tb[k, 0] - pc[0]) / vs[0] / f(osf))
            cy = f(f(tb[k, 1] - pc[1]) / vs[1] / f(osf))
            ix, iy = int(np.trunc(cx)), int(np.trunc(cy))
            if not (0 <= ix < fm[0] and 0 <= iy < fm[1]):
                continue
            draw_heatmap_gaussian(hm[cls_id], (ix, iy), radius)
            ind[k] = iy * int(fm[0]) + ix
            mk[k] = 1
            dim = np.log(tb[k, 3:6]) if head.norm_bbox else tb[k, 3:6]
            ab[k] = np.concatenate([[cx - f(ix), cy - f(iy)], [tb[k, 2]], dim,
                                    [np.sin(tb[k, 6]), np.cos(tb[k, 6])], tb[k, 7:9]]).astype(f)
        heatmaps.append(hm); anno_boxes.append(ab); inds.append(ind); masks.append(mk)
    return heatmaps, anno_boxes, inds, masks


def get_targets(head, gt_bboxes_3d, gt_labels_3d, device):
    """:366-413 -> per task stacked tensors on `device` (one upload per tensor kind)."""
    per_sample = []
    for boxes, labels in zip(gt_bboxes_3d, gt_labels_3d):
        b9 = torch.cat((boxes.gravity_center, boxes.tensor[:, 3:]), dim=1).cpu().numpy().astype(np.float32)
        lab = labels.cpu().numpy() if torch.is_tensor(labels) else np.asarray(labels)
        per_sample.append(get_targets_single_np(head, b9, lab))
    nt = len(head.task_heads)
    out = []
    for kind in range(4):
        out.append([torch.from_numpy(np.stack([s[kind][t] for s in per_sample])).to(device) for t in range(nt)])
    return tuple(out)
