"""LiDAR points -> per-camera depth maps: Python mirror of the loader transform ``PointToMultiViewDepth``
(``mmdet3d/datasets/pipelines/loading.py:18-61``) on ``dbev_points_to_depth_maps`` (csrc/depth_map.hip).

The reference runs this per sample on the data-loader's CPU (six matmuls over 240 k points + an argsort per camera) and
appends the result to ``results['img_inputs']``; here the sweep and the camera matrices are already on the device, and the
maps of all cameras (of all frames: any leading number of views) come out of one launch."""
import torch

from . import _lib as L


def camera_matrices(rots, trans, intrins, post_rots, post_trans):
    """-> f32[N, 24] per view: inverse(rots @ inverse(intrins)) [9], post_rots [9], trans [3], post_trans [3] (:50-55)"""
    combine = rots.float().matmul(torch.inverse(intrins.float()))
    cinv = torch.inverse(combine)
    n = rots.shape[0]
    return torch.cat([cinv.reshape(n, 9), post_rots.float().reshape(n, 9), trans.float().reshape(n, 3),
                      post_trans.float().reshape(n, 3)], 1).contiguous()


def points_to_depth_maps(points, rots, trans, intrins, post_rots, post_trans, height, width, downsample, dbound):
    """points f32[n, >=3] (lidar frame), camera tensors [N, ...] -> depth maps f32[N, height // downsample, width // downsample]"""
    dev = L.require_cuda(points, rots)
    pts = points.float().contiguous()
    mats = camera_matrices(rots, trans, intrins, post_rots, post_trans).to(dev)
    n_cam = mats.shape[0]
    out = torch.empty((n_cam, height // downsample, width // downsample), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        L.call("dbev_points_to_depth_maps", L.ptr(pts), pts.shape[0], pts.shape[1], L.ptr(mats), n_cam, int(height), int(width),
               int(downsample), float(dbound[0]), float(dbound[1]), L.ptr(out), L.stream_ptr(dev))
    return out


class PointToMultiViewDepth(object):
    """loading.py:18-61, same constructor and ``__call__(results)`` contract (``results['points']`` with a ``.tensor`` or a plain
    tensor, ``results['img_inputs'][:6]`` = imgs, rots, trans, intrins, post_rots, post_trans)."""

    def __init__(self, grid_config, downsample=16):
        self.downsample = downsample
        self.grid_config = grid_config

    def __call__(self, results):
        pts = results["points"]
        pts = pts.tensor if hasattr(pts, "tensor") else pts
        imgs, rots, trans, intrins, post_rots, post_trans = results["img_inputs"][:6]
        depth = points_to_depth_maps(pts, rots, trans, intrins, post_rots, post_trans, imgs.shape[2], imgs.shape[3],
                                     self.downsample, self.grid_config["dbound"])
        results["img_inputs"] = tuple(results["img_inputs"]) + (depth,)
        return results
