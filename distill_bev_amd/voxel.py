"""Voxel ops -- host-side mirror of ``mmdet3d/ops/voxel`` (voxelize.py, scatter_points.py).

Same public names and semantics as the reference:
  ``voxelization(points, voxel_size, coors_range, max_points=35, max_voxels=20000, deterministic=True)``
  ``Voxelization(voxel_size, point_cloud_range, max_num_points, max_voxels, deterministic)``
  ``dynamic_scatter(feats, coors, reduce_type)``, ``DynamicScatter(voxel_size, point_cloud_range, average_points)``
backed by the gfx950 C ABI (include/dbev_hip.h).  Results equal the reference's CPU
implementation bit for bit for every integer output; no CPU fallback.
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.nn.modules.utils import _pair

from . import _lib as L

__all__ = ["voxelization", "Voxelization", "dynamic_scatter", "DynamicScatter",
           "dynamic_point_to_voxel_forward", "dynamic_point_to_voxel_backward"]

_REDUCE = {"sum": 0, "mean": 1, "max": 2}


def _ws(nbytes, dev):
    return torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=dev)


def dynamic_voxelize(points, coors, voxel_size, coors_range, NDim=3):
    """ext ``dynamic_voxelize`` (voxelization.h:86-95): fills caller-allocated coors in place."""
    dev = L.require_cuda(points, coors)
    assert points.dtype == torch.float32 and coors.dtype == torch.int32
    assert points.is_contiguous() and coors.is_contiguous()
    with torch.cuda.device(dev):
        L.call("dbev_dynamic_voxelize", L.ptr(points), L.ptr(coors), points.size(0), points.size(1),
               L.host_floats(voxel_size), L.host_floats(coors_range), int(NDim), L.stream_ptr(dev))


def hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range,
                  max_points, max_voxels, NDim=3, deterministic=True):
    """ext ``hard_voxelize`` (voxelization.h:58-84): fills the caller-allocated outputs and
    returns voxel_num (host int, like the reference -- one device->host read)."""
    dev = L.require_cuda(points, voxels, coors, num_points_per_voxel)
    assert points.dtype == torch.float32 and points.is_contiguous()
    vs, rg = L.host_floats(voxel_size), L.host_floats(coors_range)
    with torch.cuda.device(dev):
        nbytes = L.call("dbev_hard_voxelize_workspace_bytes", points.size(0), vs, rg)
        if nbytes == 0:
            raise L.DbevHipError("hard_voxelize: invalid voxel_size / coors_range")
        ws = _ws(nbytes, dev)
        vnum = torch.zeros((1,), dtype=torch.int32, device=dev)
        L.call("dbev_hard_voxelize", L.ptr(points), L.ptr(voxels), L.ptr(coors),
               L.ptr(num_points_per_voxel), L.ptr(vnum), points.size(0), points.size(1), vs, rg,
               int(max_points), int(max_voxels), int(NDim), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    return int(vnum.item())


class _Voxelization(Function):
    """voxelize.py:10-70."""

    @staticmethod
    def forward(ctx, points, voxel_size, coors_range, max_points=35, max_voxels=20000,
                deterministic=True):
        points = points.contiguous()
        if max_points == -1 or max_voxels == -1:
            coors = points.new_zeros(size=(points.size(0), 3), dtype=torch.int)
            dynamic_voxelize(points, coors, voxel_size, coors_range, 3)
            return coors
        voxels = points.new_zeros(size=(max_voxels, max_points, points.size(1)))
        coors = points.new_zeros(size=(max_voxels, 3), dtype=torch.int)
        num_points_per_voxel = points.new_zeros(size=(max_voxels,), dtype=torch.int)
        voxel_num = hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size,
                                  coors_range, max_points, max_voxels, 3, deterministic)
        return voxels[:voxel_num], coors[:voxel_num], num_points_per_voxel[:voxel_num]


voxelization = _Voxelization.apply


class Voxelization(nn.Module):
    """voxelize.py:76-149."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000,
                 deterministic=True):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.max_num_points = max_num_points
        self.max_voxels = max_voxels if isinstance(max_voxels, tuple) else _pair(max_voxels)
        self.deterministic = deterministic
        pcr = torch.tensor(point_cloud_range, dtype=torch.float32)
        vs = torch.tensor(voxel_size, dtype=torch.float32)
        grid_size = torch.round((pcr[3:] - pcr[:3]) / vs).long()
        self.grid_size = grid_size
        self.pcd_shape = [*grid_size[:2], 1][::-1]

    def forward(self, input):
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        return voxelization(input, self.voxel_size, self.point_cloud_range, self.max_num_points,
                            max_voxels, self.deterministic)

    def __repr__(self):
        return (f"{self.__class__.__name__}(voxel_size={self.voxel_size}, point_cloud_range="
                f"{self.point_cloud_range}, max_num_points={self.max_num_points}, max_voxels="
                f"{self.max_voxels}, deterministic={self.deterministic})")


# --------------------------------------------------------------------------------------
def dynamic_scatter_prepare(coors, grid=None):
    """Grouping half of ``dynamic_point_to_voxel_forward``: returns a dict with
    out_coors[M,3], coors_map[N], reduce_count[M], vstart[N+1], vlist[N], M.
    ``grid`` = (gz, gy, gx) bound on the coordinates; derived from the data (one sync)
    when not given.  The result can be shared by several reductions over the same coors
    (DynamicPillarFeatureNet calls the op twice with identical coors,
    pillar_encoder.py:304,331)."""
    dev = L.require_cuda(coors)
    coors = coors.contiguous()
    assert coors.dtype == torch.int32 and coors.dim() == 2 and coors.size(1) == 3
    n = coors.size(0)
    if n == 0:
        z = coors.new_empty((0,), dtype=torch.int32)
        return dict(out_coors=coors.new_empty((0, 3)), coors_map=z, reduce_count=z,
                    vstart=coors.new_zeros((1,)), vlist=z, M=0, N=0)
    if grid is None:
        mx = coors.max(dim=0).values.clamp(min=0).tolist()
        grid = (int(mx[0]) + 1, int(mx[1]) + 1, int(mx[2]) + 1)
    gz, gy, gx = (int(g) for g in grid)
    with torch.cuda.device(dev):
        nbytes = L.call("dbev_dynamic_scatter_workspace_bytes", n, gz, gy, gx)
        if nbytes == 0:
            raise L.DbevHipError("dynamic_scatter: invalid grid")
        ws = _ws(nbytes, dev)
        out_coors = torch.empty((n, 3), dtype=torch.int32, device=dev)
        cmap = torch.empty((n,), dtype=torch.int32, device=dev)
        cnt = torch.empty((n,), dtype=torch.int32, device=dev)
        vstart = torch.empty((n + 1,), dtype=torch.int32, device=dev)
        vlist = torch.empty((n,), dtype=torch.int32, device=dev)
        mdev = torch.empty((1,), dtype=torch.int32, device=dev)
        L.call("dbev_dynamic_scatter_prepare", L.ptr(coors), n, gz, gy, gx, L.ptr(out_coors),
               L.ptr(cmap), L.ptr(cnt), L.ptr(vstart), L.ptr(vlist), L.ptr(mdev), L.ptr(ws),
               ws.numel(), L.stream_ptr(dev))
    M = int(mdev.item())
    return dict(out_coors=out_coors[:M], coors_map=cmap, reduce_count=cnt[:M], vstart=vstart,
                vlist=vlist, M=M, N=n)


def dynamic_scatter_reduce(feats, prep, reduce_type):
    dev = L.require_cuda(feats)
    feats = feats.contiguous()
    assert feats.dtype == torch.float32
    M, C = prep["M"], feats.size(1)
    reduced = torch.empty((M, C), dtype=feats.dtype, device=dev)
    if M > 0:
        with torch.cuda.device(dev):
            L.call("dbev_dynamic_scatter_reduce", L.ptr(feats), L.ptr(prep["vstart"]),
                   L.ptr(prep["vlist"]), L.ptr(reduced), M, C, _REDUCE[reduce_type],
                   L.stream_ptr(dev))
    return reduced


def dynamic_point_to_voxel_forward(feats, coors, reduce_type, grid=None):
    """ext fn (voxelization.h:108-121) -> [reduced_feats, out_coors, coors_map, reduce_count]
    (+ the CSR kept for the backward as 5th/6th element)."""
    if feats.size(0) == 0:  # scatter_points_cuda.cu:192-196
        e = coors.new_empty((0,), dtype=torch.int32)
        return [feats.clone().detach(), coors.clone().detach(), e, e, coors.new_zeros((1,)), e]
    prep = dynamic_scatter_prepare(coors, grid)
    reduced = dynamic_scatter_reduce(feats, prep, reduce_type)
    return [reduced, prep["out_coors"], prep["coors_map"], prep["reduce_count"], prep["vstart"],
            prep["vlist"]]


def dynamic_point_to_voxel_backward(grad_feats, grad_reduced_feats, feats, reduced_feats,
                                    coors_idx, reduce_count, reduce_type, vstart, vlist):
    """ext fn (voxelization.h:123-140); writes into caller-allocated grad_feats."""
    dev = L.require_cuda(grad_feats, grad_reduced_feats, feats, reduced_feats)
    n, C = feats.shape
    if n == 0:
        return
    with torch.cuda.device(dev):
        L.call("dbev_dynamic_scatter_backward", L.ptr(grad_feats), L.ptr(grad_reduced_feats),
               L.ptr(feats), L.ptr(reduced_feats), L.ptr(coors_idx), L.ptr(reduce_count),
               L.ptr(vstart), L.ptr(vlist), n, reduced_feats.size(0), C, _REDUCE[reduce_type],
               L.stream_ptr(dev))


class _dynamic_scatter(Function):
    """scatter_points.py:9-47."""

    @staticmethod
    def forward(ctx, feats, coors, reduce_type="max", grid=None, prep=None):
        feats = feats.contiguous()
        if prep is None:
            res = dynamic_point_to_voxel_forward(feats, coors, reduce_type, grid)
        else:
            res = [dynamic_scatter_reduce(feats, prep, reduce_type), prep["out_coors"],
                   prep["coors_map"], prep["reduce_count"], prep["vstart"], prep["vlist"]]
        voxel_feats, voxel_coors, point2voxel_map, voxel_points_count, vstart, vlist = res
        ctx.reduce_type = reduce_type
        ctx.save_for_backward(feats, voxel_feats, point2voxel_map, voxel_points_count, vstart, vlist)
        ctx.mark_non_differentiable(voxel_coors)
        return voxel_feats, voxel_coors

    @staticmethod
    def backward(ctx, grad_voxel_feats, grad_voxel_coors=None):
        feats, voxel_feats, point2voxel_map, voxel_points_count, vstart, vlist = ctx.saved_tensors
        grad_feats = torch.empty_like(feats)
        dynamic_point_to_voxel_backward(grad_feats, grad_voxel_feats.contiguous(), feats, voxel_feats,
                                        point2voxel_map, voxel_points_count, ctx.reduce_type,
                                        vstart, vlist)
        return grad_feats, None, None, None, None


def dynamic_scatter(feats, coors, reduce_type="max", grid=None, prep=None):
    return _dynamic_scatter.apply(feats, coors, reduce_type, grid, prep)


class DynamicScatter(nn.Module):
    """scatter_points.py:53-122: per-sample loop when coors carries a batch column."""

    def __init__(self, voxel_size, point_cloud_range, average_points: bool):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.average_points = average_points
        pcr = torch.tensor(point_cloud_range, dtype=torch.float32)
        vs = torch.tensor(voxel_size, dtype=torch.float32)
        g = torch.round((pcr[3:] - pcr[:3]) / vs).long().tolist()  # x, y, z cells
        self.grid = (max(int(g[2]), 1), int(g[1]), int(g[0]))       # (gz, gy, gx)

    def forward_single(self, points, coors, prep=None):
        reduce = "mean" if self.average_points else "max"
        # grid=None: bounds derived from the data (the reference op is grid-agnostic);
        # callers that KNOW the coors come from this grid pass prep/grid themselves.
        return dynamic_scatter(points.contiguous(), coors.contiguous(), reduce, None, prep)

    def forward(self, points, coors):
        if coors.size(-1) == 3:
            return self.forward_single(points, coors)
        batch_size = int(coors[-1, 0] + 1)
        voxels, voxel_coors = [], []
        for i in range(batch_size):
            inds = torch.where(coors[:, 0] == i)
            voxel, voxel_coor = self.forward_single(points[inds], coors[inds][:, 1:])
            coor_pad = nn.functional.pad(voxel_coor, (1, 0), mode="constant", value=i)
            voxel_coors.append(coor_pad)
            voxels.append(voxel)
        return torch.cat(voxels, dim=0), torch.cat(voxel_coors, dim=0)

    def __repr__(self):
        return (f"{self.__class__.__name__}(voxel_size={self.voxel_size}, point_cloud_range="
                f"{self.point_cloud_range}, average_points={self.average_points})")
