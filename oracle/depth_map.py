"""Oracle (CPU, numpy) for the LiDAR -> multi-view depth maps -- TEST INFRASTRUCTURE ONLY.

Restates PointToMultiViewDepth (mmdet3d/datasets/pipelines/loading.py:18-61): project the sweep into every camera
(:50-55), round to the down-sampled pixel grid, keep points inside the map with dbound[0] <= depth < dbound[1] (:26-31),
and give every pixel the nearest point that lands on it.  The reference obtains "nearest" by sorting on the float32 key
pixel_rank + depth / 100 and keeping the first entry of each pixel (:33-41); this restatement takes the minimum directly
(fp64 projection optional, for telling rounding flips from logic errors).  Pinned against the imported loading.py by
tests/test_oracle_depth_map.py (fixture depth_map.npz)."""
import numpy as np


def points_to_depth_maps(points, rots, trans, intrins, post_rots, post_trans, height, width, downsample, dbound, dtype=np.float32):
    pts = np.asarray(points, dtype)[:, :3]
    h, w = height // downsample, width // downsample
    out = np.zeros((rots.shape[0], h, w), np.float32)
    for c in range(rots.shape[0]):
        combine = np.asarray(rots[c], dtype) @ np.linalg.inv(np.asarray(intrins[c], dtype))
        cinv = np.linalg.inv(combine).astype(dtype)
        p = (pts - np.asarray(trans[c], dtype)[None]) @ cinv.T
        p = np.concatenate([p[:, :2] / p[:, 2:3], p[:, 2:3]], 1)
        p = p @ np.asarray(post_rots[c], dtype).T + np.asarray(post_trans[c], dtype)[None]
        coor = np.rint(p[:, :2] / dtype(downsample))                   # half to even, as torch.round
        d = p[:, 2]
        keep = (coor[:, 0] >= 0) & (coor[:, 0] < w) & (coor[:, 1] >= 0) & (coor[:, 1] < h) & (d < dbound[1]) & (d >= dbound[0])
        cx, cy, d = coor[keep, 0].astype(np.int64), coor[keep, 1].astype(np.int64), d[keep].astype(np.float32)
        m = np.full((h, w), np.inf, np.float32)
        np.minimum.at(m, (cy, cx), d)
        m[np.isinf(m)] = 0.0
        out[c] = m
    return out
