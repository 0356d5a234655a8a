"""TEST INFRASTRUCTURE (never imported by the product): numpy restatement of the voxel teachers' dynamic voxel encoders,
mmdet3d/models/voxel_encoders/dynamic_voxel_encoder.py -- `voxelization` :8-17, `voxelization_virtual` :19-68,
`DynamicVoxelEncoder.forward` :83-102 -- with mmdet3d/core/utils/scatter.py:37-60 (`scatter_mean` = sequential
`scatter_add_` / count).  Pinned by tests/golden/dynvoxel.npz, which the imported reference file computed
(tests/golden/make_golden.py dynvoxel; tests/test_oracle_dynvoxel.py).

Formulated per voxel instead of per padded row: the reference sums a zero-padded [N, 24] copy of the cloud whose rows are
reordered real -> painted -> virtual; real rows only touch columns 0..5 and 23, painted / virtual rows only 6..22, so per
column the summation order is "real rows in input order" resp. "painted rows in input order, then virtual rows in input order"."""
import numpy as np


def _coords(points, pc_range, voxel_size):
    pc_range = np.asarray(pc_range, np.float32)
    voxel_size = np.asarray(voxel_size, np.float32)
    keep = np.ones(len(points), bool)
    for k in range(3):                                                          # :9-11 closed range
        keep &= (points[:, k] >= pc_range[k]) & (points[:, k] <= pc_range[3 + k])
    q = (points[:, [2, 1, 0]] - pc_range[[2, 1, 0]]) / voxel_size[[2, 1, 0]]    # fp32, :13
    return keep, np.trunc(q).astype(np.int64)


def _unique_rows(coords):
    """coords.unique(dim=0, return_inverse=True): lexicographically sorted rows"""
    uc, inv = np.unique(coords, axis=0, return_inverse=True)
    return uc, inv.reshape(-1)


def _seq_sum(rows, inv, m):
    out = np.zeros((m, rows.shape[1]), np.float32)
    np.add.at(out, inv, rows)                                                   # unbuffered, in row order
    return out


def voxelization(points, pc_range, voxel_size):
    points = np.asarray(points, np.float32)
    keep, coords = _coords(points, pc_range, voxel_size)
    points, coords = points[keep], coords[keep]
    uc, inv = _unique_rows(coords)
    cnt = np.maximum(np.bincount(inv, minlength=len(uc)), 1).astype(np.float32)
    return _seq_sum(points, inv, len(uc)) / cnt[:, None], uc


def voxelization_virtual(points, pc_range, voxel_size):
    points = np.asarray(points, np.float32)
    assert points.shape[1] == 17
    keep, coords = _coords(points, pc_range, voxel_size)
    points, coords = points[keep], coords[keep]
    tag = points[:, -2]
    assert np.isin(tag, (1.0, 0.0, -1.0)).all(), "the reference cannot represent other tags (row counts would not add up)"
    uc, inv = _unique_rows(coords)
    m = len(uc)
    cnt = np.maximum(np.bincount(inv, minlength=m), 1).astype(np.float32)
    real, painted, virtual = tag == 1, tag == 0, tag == -1
    s = np.zeros((m, 24), np.float32)
    s[:, :6] = _seq_sum(points[real][:, [0, 1, 2, 3, 4, 16]], inv[real], m)
    s[:, 23] = np.bincount(inv[real], minlength=m)
    pv_rows = np.concatenate([points[painted], points[virtual]])
    pv_inv = np.concatenate([inv[painted], inv[virtual]])
    s[:, 6:22] = _seq_sum(pv_rows[:, :16], pv_inv, m)                           # 15 columns + the tag column
    s[:, 22] = np.bincount(inv[painted], minlength=m)
    mean = s / cnt[:, None]
    ind = mean[:, 23]
    mix = (ind > 0) & (ind < 1)
    vox = mean[:, :23].copy()
    vox[mix, :6] = vox[mix, :6] / ind[mix, None]
    vox[mix, 6:] = vox[mix, 6:] / (np.float32(1) - ind[mix, None])
    return vox, uc


def dynamic_voxel_encoder(points_list, pc_range, voxel_size, virtual=False):
    fn = voxelization_virtual if virtual else voxelization
    vox, co = [], []
    for b, p in enumerate(points_list):
        v, c = fn(p, pc_range, voxel_size)
        vox.append(v)
        co.append(np.concatenate([np.full((len(c), 1), b, np.int64), c], 1))
    pr, vs = np.asarray(pc_range, np.float32), np.asarray(voxel_size, np.float32)
    return np.concatenate(vox), np.concatenate(co), np.round((pr[3:] - pr[:3]) / vs).astype(np.int32)
