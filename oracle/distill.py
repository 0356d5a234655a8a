"""Oracle (CPU, numpy) for the foreground-masked BEV-feature distillation loss.
TEST INFRASTRUCTURE ONLY.

Restates mmdet3d/models/detectors/bevdet_distill.py (foreground_scale_mask :755-843,
add_fp_as_fg :846-970 mode 'teacher' / fp_scale_mode 'average', fgd_distill_loss
:1084-1108,1163-1168,1253-1287) and the numpy geometry it calls
(mmdet3d/core/bbox/box_np_ops.py: corners_nd :49-80, rotation_3d_in_axis :175-203,
center_to_corner_box3d :206-235, corner_to_surfaces_3d :404-423, surface_equ_3d :694-715,
points_in_rbbox :426-446, _points_in_convex_polygon_3d_jit :719-753).

Pinning: the geometry half (points_in_rbbox -> fg / fg_scale / bg_scale maps) is pinned
against the imported reference box_np_ops (tests/golden/fgmask_*.npz); the loss half (attention
masks, combine_gt scaling, fp re-scaling, the three masked sums, the spatial term, the adaptation
layers) against losses computed by the reference's own bevdet_distill.py imported by path
(BEVDetDistill.fgd_distill_loss on a bare instance: tests/golden/fgd_losses.npz, make_golden.py
'fgd'; tests/test_oracle_step_ops.py).
"""
import numpy as np

f32 = np.float32


# ---- box geometry (float32, like the reference's numpy arrays) -------------------------
def box_corners(boxes):
    """boxes f32[M, 7] = (x, y, z_bottom, w, l, h, yaw) -> corners f32[M, 8, 3].
    center_to_corner_box3d(origin=(0.5, 0.5, 0), axis=2): unit-cube corner pattern
    (0,0,0),(0,0,1),(0,1,1),(0,1,0),(1,0,0),(1,0,1),(1,1,1),(1,1,0) minus origin, scaled by
    dims, rotated about z by yaw with rot_mat_T = [[c,-s,0],[s,c,0],[0,0,1]] applied as
    einsum('aij,jka->aik'), plus centre."""
    boxes = np.asarray(boxes)
    dt = boxes.dtype
    pat = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 1], [0, 1, 0],
                    [1, 0, 0], [1, 0, 1], [1, 1, 1], [1, 1, 0]], dtype=dt)
    pat = pat - np.array([0.5, 0.5, 0.0], dtype=dt)
    corners = boxes[:, None, 3:6] * pat[None]                       # [M,8,3]
    s, c = np.sin(boxes[:, 6]), np.cos(boxes[:, 6])
    z, o = np.zeros_like(c), np.ones_like(c)
    rot_t = np.stack([[c, -s, z], [s, c, z], [z, z, o]])            # [3(j),3(k),M]
    corners = np.einsum("aij,jka->aik", corners, rot_t)
    return corners + boxes[:, None, :3]


_SURF = np.array([[0, 1, 2, 3], [7, 6, 5, 4], [0, 3, 7, 4], [1, 5, 6, 2], [0, 4, 5, 1], [3, 2, 6, 7]])


def box_planes(boxes):
    """-> (normal f32[M,6,3], d f32[M,6]) with  n.p + d < 0  <=> p strictly inside
    (surface_equ_3d: n = cross(p0-p1, p1-p2), d = -n.p0 over the 6 inward-facing faces)."""
    corners = box_corners(boxes)
    surf = corners[:, _SURF]                                        # [M,6,4,3]
    vec = surf[:, :, :2] - surf[:, :, 1:3]
    normal = np.cross(vec[:, :, 0], vec[:, :, 1])
    d = np.einsum("aij,aij->ai", normal, surf[:, :, 0])
    return normal, -d


def points_in_rbbox(points, boxes):
    """-> bool[N, M]; a point is inside iff all six plane signs are < 0, each sign evaluated
    as ((px*nx + py*ny) + pz*nz) + d in the arrays' dtype (box_np_ops.py:746-752)."""
    normal, d = box_planes(boxes)
    p = np.asarray(points)[:, :3]
    sign = (p[:, None, None, 0] * normal[None, :, :, 0] + p[:, None, None, 1] * normal[None, :, :, 1]
            + p[:, None, None, 2] * normal[None, :, :, 2] + d[None])
    return np.all(sign < 0, axis=2)


def cell_coords(n, voxel, osf, lo):
    """bevdet_distill.py:766-767: [i * voxel_size * out_size_factor + pc_range for i] evaluated
    with 0-dim float32 torch tensors -> float32 (i*vs rounded, *osf rounded, +lo rounded)."""
    out = np.empty(n, dtype=f32)
    for i in range(n):
        out[i] = f32(f32(f32(i) * f32(voxel)) * f32(osf)) + f32(lo)
    return out


def foreground_scale_mask(H, W, boxes_list, grid_size=(1024, 1024, 40),
                          pc_range=(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0),
                          voxel_size=(0.1, 0.1, 0.2)):
    """bevdet_distill.py:755-843 (transpose_mask=False, no bg_extend, avg_fg_scale_mask off).
    boxes_list: per sample f32[M_b, >=7] LiDAR boxes (x,y,z_bottom,w,l,h,yaw,...).
    -> (fg f32[B,1,H,W], fg_scale f32[B,1,H,W], bg_scale f32[B,1,H,W]);  [.., iy, ix]."""
    assert H == W and grid_size[0] % W == 0
    osf = grid_size[0] // W
    xs = cell_coords(W, voxel_size[0], osf, pc_range[0])
    ys = cell_coords(H, voxel_size[1], osf, pc_range[1])
    gx, gy = np.meshgrid(xs, ys, indexing="ij")
    pts = np.stack([gx.reshape(-1), gy.reshape(-1), np.full(H * W, 0.5, dtype=f32)], 1).astype(f32)
    area = f32(f32(f32(voxel_size[0]) * f32(voxel_size[1])) * f32(osf)) * f32(osf)
    fgs, fss, bss = [], [], []
    for boxes in boxes_list:
        b = np.array(boxes, dtype=f32)[:, :7].copy()
        b[:, 2] = 0
        b[:, 5] = 1
        fg = np.zeros(H * W, dtype=np.float64)
        fs = np.zeros(H * W, dtype=np.float64)
        if b.shape[0] > 0:
            m = points_in_rbbox(pts, b)
            fg = m.any(1).astype(np.float64)
            first = m.argmax(1)                                     # lowest box index that hits
            hit = m.any(1)
            val = np.sqrt((area / (b[first, 3] * b[first, 4]).astype(f32)).astype(f32)).astype(f32)
            fs[hit] = val[hit]
        bs = np.full(H * W, 1.0 / (H * W - np.sum(fg != 0)), dtype=np.float64)
        fgs.append(fg.reshape(W, H).T.reshape(1, 1, H, W))
        fss.append(fs.reshape(W, H).T.reshape(1, 1, H, W))
        bss.append(bs.reshape(W, H).T.reshape(1, 1, H, W))
    return (np.concatenate(fgs).astype(f32), np.concatenate(fss).astype(f32),
            np.concatenate(bss).astype(f32))


# ---- loss arithmetic (fp64 yardstick) ---------------------------------------------------
def _softmax(x, axis):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def attention_masks(teacher, student, spatial_t=0.5, channel_t=0.5, s_ratio=1.0, mode="teacher_student"):
    """bevdet_distill.py:1084-1108.  -> (sum_att [B,1,H,W], c_att [B,C,1,1]) fp64."""
    T = teacher.astype(np.float64)
    S = student.astype(np.float64)
    B, C, H, W = T.shape
    t_att = _softmax(np.abs(T).mean(1).reshape(B, -1) / spatial_t, 1) * (H * W)
    s_att = _softmax(np.abs(S).mean(1).reshape(B, -1) / spatial_t, 1) * (H * W)
    c_att = _softmax(np.abs(T).mean((2, 3)) / channel_t, 1) * C
    if mode == "teacher":
        att = t_att
    else:
        att = (t_att + s_att * s_ratio) / (1 + s_ratio)
    return att.reshape(B, 1, H, W), c_att.reshape(B, C, 1, 1)


def fp_masks(fg, gt_hm_max, teacher_hm_max, thres=0.1):
    """add_fp_as_fg mode 'teacher', fp_scale_mode 'average' (:891-892, :914-920), same-size maps.
    -> (fp [B,1,H,W], fp_scale, n_fp [B])."""
    fp = ((gt_hm_max < thres) & (teacher_hm_max > thres) & (fg == 0)).astype(np.float64)
    n = fp.sum((1, 2, 3))
    sc = np.zeros_like(fp)
    for b in range(fp.shape[0]):
        if n[b] > 0:
            sc[b][fp[b] > 0] = 1.0 / n[b]
    return fp, sc, n


def fgd_feature_losses(student, teacher, fg, fg_scale, bg_scale, fp=None, fp_scale=None, n_fp=None,
                       w_fg=6e-3, w_bg=4e-2, w_fp=6e-2, spatial_t=0.5, channel_t=0.5,
                       spatial_att="teacher_student", channel_mask=False):
    """kd_fg / kd_bg / kd_fp of fgd_distill_loss (:1110-1129 bg re-scale, :1163-1168 combine_gt,
    :1253-1262, :1282-1287) in fp64.  student = adapted student features."""
    S = student.astype(np.float64)
    T = teacher.astype(np.float64)
    B, C, H, W = S.shape
    att, c_att = attention_masks(teacher, student, spatial_t, channel_t, 1.0, spatial_att)
    fgm = fg.astype(np.float64)
    bgm = (fgm == 0).astype(np.float64)
    bgs = bg_scale.astype(np.float64).copy()
    if fp is not None:
        bgm[fp != 0] = 0
        n_bg = H * W - fgm.sum((1, 2, 3))
        for b in range(B):
            bgs[b][:] = 1.0 / (n_bg[b] - n_fp[b]) if n_bg[b] > n_fp[b] else 0.0
    scale = np.maximum(fg_scale.astype(np.float64), bgs)
    fg_w = fgm * scale * att
    bg_w = bgm * scale * att
    if channel_mask:
        fg_w = fg_w * c_att
        bg_w = bg_w * c_att
    sq = (S - T) ** 2
    out = {"kd_fg_feat_loss": (sq * fg_w).sum() * w_fg / B,
           "kd_bg_feat_loss": (sq * bg_w).sum() * w_bg / B}
    if fp is not None:
        fpw = fp * fp_scale * att * c_att
        out["kd_fp_bg_feat_loss"] = (sq * fpw).sum() * w_fp / B
    return out, dict(att=att, c_att=c_att, fg_w=fg_w, bg_w=bg_w, scale=scale)


# ---- adaptation layers + spatial term (fp64 yardstick) ----------------------------------------------
def conv1x1(x, w, b=None):
    """nn.Conv2d(k=1): x [B,Ci,H,W], w [Co,Ci,1,1] -> [B,Co,H,W] (bevdet_distill.py:232-234)."""
    y = np.einsum("bchw,oc->bohw", x.astype(np.float64), w.reshape(w.shape[0], -1).astype(np.float64))
    return y if b is None else y + b.astype(np.float64).reshape(1, -1, 1, 1)


def upsample_bilinear_ac(x, s):
    """nn.Upsample(scale_factor=s, mode='bilinear', align_corners=True) (:275-277)."""
    B, C, H, W = x.shape
    OH, OW = H * s, W * s
    x = x.astype(np.float64)

    def taps(n_in, n_out):
        pos = np.arange(n_out, dtype=np.float64) * ((n_in - 1) / (n_out - 1) if n_out > 1 else 0.0)
        i0 = np.minimum(np.floor(pos).astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        return i0, i1, pos - i0
    y0, y1, fy = taps(H, OH)
    x0, x1, fx = taps(W, OW)
    top = x[:, :, y0][:, :, :, x0] * (1 - fx) + x[:, :, y0][:, :, :, x1] * fx
    bot = x[:, :, y1][:, :, :, x0] * (1 - fx) + x[:, :, y1][:, :, :, x1] * fx
    return top * (1 - fy)[None, None, :, None] + bot * fy[None, None, :, None]


def bn_train_relu(x, gamma, beta, eps=1e-5):
    """training-mode BatchNorm2d (biased batch variance) + ReLU"""
    m = x.mean((0, 2, 3), keepdims=True)
    v = x.var((0, 2, 3), keepdims=True)
    y = (x - m) / np.sqrt(v + eps) * gamma.astype(np.float64).reshape(1, -1, 1, 1) + beta.astype(np.float64).reshape(1, -1, 1, 1)
    return np.maximum(y, 0)


def three_layer(x, sd, prefix=""):
    """ThreeLayer (bevdet_distill.py:99-132), kernel 1 / stride 1: 3 x (1x1 conv -> BN(train) -> ReLU).
    sd: mapping name -> array with keys '<prefix>conv1__weight' ... as stored by the golden generator."""
    for i in (1, 2, 3):
        x = conv1x1(x, sd[f"{prefix}conv{i}__weight"], sd[f"{prefix}conv{i}__bias"])
        x = bn_train_relu(x, sd[f"{prefix}norm{i}__weight"], sd[f"{prefix}norm{i}__bias"])
    return x


def conv3x3_single(x, w, b):
    """nn.Conv2d(1, 1, 3, padding=1) (spatial_wise_adaptations, :345-348): x [B,1,H,W]."""
    B, _, H, W = x.shape
    xp = np.zeros((B, H + 2, W + 2))
    xp[:, 1:-1, 1:-1] = x[:, 0]
    y = np.zeros((B, H, W))
    for i in range(3):
        for j in range(3):
            y += float(w[0, 0, i, j]) * xp[:, i:i + H, j:j + W]
    return (y + float(b[0]))[:, None]


def spatial_loss(teacher, student, w, b, weight):
    """kd_spatial_loss (:1272-1278): sum |mean_c T - conv3x3(mean_c S)| * weight / B."""
    T = teacher.astype(np.float64).mean(1, keepdims=True)
    S = conv3x3_single(student.astype(np.float64).mean(1, keepdims=True), w, b)
    return np.abs(T - S).sum() * weight / teacher.shape[0]
