"""Seeded synthetic nuScenes-shaped inputs (SURVEY.md 8(d) / BASELINE.md 3).

No dataset is reachable from the build or GPU boxes, so every test, the smoke
run and bench.py draw their inputs from here.  Shapes and value ranges follow
the reference's input contract
(``mmdet3d/datasets/pipelines/loading.py:243-387`` img_inputs 7-tuple,
``:654`` multi-sweep points, ``nuscenes_dataset.py`` 9-dof boxes).
Pure numpy so that the oracle, the fixtures and the HIP path see identical
bits; callers move the arrays to the device.
"""
import math

import numpy as np

CAM_YAWS_DEG = (55.0, 0.0, -55.0, 110.0, 180.0, -110.0)


def _rz(yaw):
    c, s = math.cos(yaw), math.sin(yaw)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def camera_rig(B, rng, n_cams=6, jitter=0.2, input_size=(256, 704), src_size=(900, 1600)):
    """6-camera surround rig.  Returns dict of f32 arrays:
    rots[B,N,3,3], trans[B,N,3], intrins[B,N,3,3], post_rots[B,N,3,3], post_trans[B,N,3]."""
    cam2ego_axes = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
    rots = np.zeros((B, n_cams, 3, 3))
    trans = np.zeros((B, n_cams, 3))
    intr = np.zeros((B, n_cams, 3, 3))
    prot = np.zeros((B, n_cams, 3, 3))
    ptr = np.zeros((B, n_cams, 3))
    K = np.array([[1266.4, 0.0, 816.3], [0.0, 1266.4, 491.5], [0.0, 0.0, 1.0]])
    s = input_size[1] / src_size[1]  # 0.44
    crop_h = int(src_size[0] * s) - input_size[0]  # 140
    for b in range(B):
        for n in range(n_cams):
            yaw = math.radians(CAM_YAWS_DEG[n % len(CAM_YAWS_DEG)])
            rots[b, n] = _rz(yaw) @ cam2ego_axes
            trans[b, n] = np.array([1.5, 0.0, 1.5]) + rng.uniform(-jitter, jitter, 3)
            intr[b, n] = K
            prot[b, n] = np.diag([s, s, 1.0])
            ptr[b, n] = np.array([0.0, -float(crop_h), 0.0])
    f = np.float32
    return dict(rots=rots.astype(f), trans=trans.astype(f), intrins=intr.astype(f),
                post_rots=prot.astype(f), post_trans=ptr.astype(f))


def lidar_points(n_points, rng, n_feats=5):
    """points f32[N, 5] = (x, y, z, intensity, dt); ~10 % outside the
    [-51.2, 51.2] x/y range and some outside z in [-5, 3]."""
    p = np.empty((n_points, n_feats), dtype=np.float32)
    p[:, 0] = rng.uniform(-54.0, 54.0, n_points)
    p[:, 1] = rng.uniform(-54.0, 54.0, n_points)
    p[:, 2] = rng.uniform(-5.5, 3.5, n_points)
    if n_feats > 3:
        p[:, 3] = rng.uniform(0.0, 255.0, n_points)
    if n_feats > 4:
        p[:, 4] = rng.integers(0, 10, n_points) * 0.05
    for k in range(5, n_feats):
        p[:, k] = rng.uniform(0.0, 1.0, n_points)
    return p


def mvp_virtual_points(n_real, n_painted, n_virtual, rng):
    """MVP multi-sweep cloud f32[n, 17] (configs/teacher_transformer/mvpformer.py:160-172, use_dim=range(17)): xyz + 14
    attribute columns; column -2 marks the kind -- 1 real LiDAR return, 0 painted, -1 virtual (dynamic_voxel_encoder.py:19-34)."""
    n = n_real + n_painted + n_virtual
    p = rng.uniform(0.0, 1.0, (n, 17)).astype(np.float32)
    p[:, 0] = rng.uniform(-54.0, 54.0, n)
    p[:, 1] = rng.uniform(-54.0, 54.0, n)
    p[:, 2] = rng.uniform(-5.5, 3.5, n)
    p[:n_real, -2] = 1.0
    p[n_real:n_real + n_painted, -2] = 0.0
    p[n_real + n_painted:, -2] = -1.0
    return p


CLASS_DIMS = np.array([  # (w, l, h) class-typical sizes, nuScenes 10 classes
    [1.95, 4.60, 1.73], [2.50, 6.90, 2.80], [2.80, 6.40, 3.20], [2.95, 11.0, 3.50],
    [2.90, 12.3, 3.90], [2.50, 0.50, 1.00], [0.77, 2.10, 1.50], [0.60, 1.70, 1.30],
    [0.67, 0.73, 1.77], [0.41, 0.41, 1.07]], dtype=np.float32)


def gt_boxes(n_boxes, rng):
    """(boxes f32[M,9] = x,y,z_bottom,w,l,h,yaw,vx,vy ; labels int64[M]) --
    the ``LiDARInstance3DBoxes.tensor`` layout (lidar_box3d.py:41-47)."""
    labels = rng.integers(0, 10, n_boxes).astype(np.int64)
    b = np.zeros((n_boxes, 9), dtype=np.float32)
    b[:, 0] = rng.uniform(-45.0, 45.0, n_boxes)
    b[:, 1] = rng.uniform(-45.0, 45.0, n_boxes)
    b[:, 2] = rng.uniform(-2.0, 0.0, n_boxes)
    b[:, 3:6] = CLASS_DIMS[labels] * rng.uniform(0.9, 1.1, (n_boxes, 1)).astype(np.float32)
    b[:, 6] = rng.uniform(-math.pi, math.pi, n_boxes)
    b[:, 7:9] = rng.uniform(-3.0, 3.0, (n_boxes, 2))
    return b, labels


def depth_gt(B, n_cams, fH, fW, rng):
    d = rng.uniform(1.0, 60.0, (B, n_cams, fH, fW)).astype(np.float32)
    m = rng.uniform(0, 1, (B, n_cams, fH, fW)) < 0.05
    return (d * m).astype(np.float32)


def lss_inputs(B, rng, n_cams=6, D=59, fH=16, fW=44, C=64):
    """Depth distribution + image features of one frame, as the view
    transformer's lift sees them: depth_prob f32[B*N, D, fH, fW] (softmaxed),
    img_feat f32[B*N, C, fH, fW]."""
    logits = rng.normal(0.0, 1.5, (B * n_cams, D, fH, fW))
    e = np.exp(logits - logits.max(1, keepdims=True))
    depth = (e / e.sum(1, keepdims=True)).astype(np.float32)
    feat = rng.normal(0.0, 1.0, (B * n_cams, C, fH, fW)).astype(np.float32)
    return depth, feat
