"""PointPillarsScatter -- host-side mirror of
``mmdet3d/models/middle_encoders/pillar_scatter.py`` on the gfx950 C ABI
(``dbev_pillars_scatter``): one pass over the whole batch canvas instead of the
reference's per-sample python loop (zero-fill + boolean-mask gather + index_put + stack).
"""
import torch
from torch import nn
from torch.autograd import Function

from . import _lib as L


class _PillarsScatter(Function):
    @staticmethod
    def forward(ctx, voxel_features, coors, batch_size, ny, nx, channels_last):
        dev = L.require_cuda(voxel_features, coors)
        vf = voxel_features.contiguous()
        assert vf.dtype == torch.float32
        co = coors.int().contiguous()
        M, C = vf.shape
        if channels_last:
            canvas = torch.empty((batch_size, C, ny, nx), dtype=vf.dtype, device=dev,
                                 memory_format=torch.channels_last)
        else:
            canvas = torch.empty((batch_size, C, ny, nx), dtype=vf.dtype, device=dev)
        cellmap = torch.empty((batch_size * ny * nx,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            L.call("dbev_pillars_scatter", L.ptr(vf), L.ptr(co), M, C, int(batch_size), int(ny), int(nx),
                   L.ptr(canvas), 1 if channels_last else 0, L.ptr(cellmap), L.stream_ptr(dev))
        ctx.save_for_backward(co, cellmap)
        ctx.dims = (M, C, int(batch_size), int(ny), int(nx), bool(channels_last))
        return canvas

    @staticmethod
    def backward(ctx, grad_canvas):
        co, cellmap = ctx.saved_tensors
        M, C, B, ny, nx, cl = ctx.dims
        dev = grad_canvas.device
        g = grad_canvas.contiguous(memory_format=torch.channels_last) if cl else grad_canvas.contiguous()
        grad_feats = torch.empty((M, C), dtype=g.dtype, device=dev)
        if M > 0:
            with torch.cuda.device(dev):
                L.call("dbev_pillars_scatter_backward", L.ptr(g), L.ptr(co), L.ptr(cellmap), M, C, B, ny,
                       nx, 1 if cl else 0, L.ptr(grad_feats), L.stream_ptr(dev))
        return grad_feats, None, None, None, None, None


def pillars_scatter(voxel_features, coors, batch_size, ny, nx, channels_last=False):
    """coors int[M, 4] = (b, z, y, x) -> f32[B, C, ny, nx] (optionally channels_last strides)."""
    return _PillarsScatter.apply(voxel_features, coors, batch_size, ny, nx, channels_last)


class PointPillarsScatter(nn.Module):
    """pillar_scatter.py:10-102 (same ctor args / call contract)."""

    def __init__(self, in_channels, output_shape, channels_last=False):
        super().__init__()
        self.output_shape = output_shape
        self.ny = output_shape[0]
        self.nx = output_shape[1]
        self.in_channels = in_channels
        self.fp16_enabled = False
        self.channels_last = channels_last

    def forward(self, voxel_features, coors, batch_size=None):
        if batch_size is not None:
            return self.forward_batch(voxel_features, coors, batch_size)
        return self.forward_single(voxel_features, coors)

    def forward_single(self, voxel_features, coors):
        """coors int[M, 3+] = (b?, y, x) per pillar_scatter.py:52 (columns 1 and 2)."""
        z = torch.zeros_like(coors[:, :1])
        co = torch.cat([z, z, coors[:, 1:2], coors[:, 2:3]], 1)
        return [pillars_scatter(voxel_features, co, 1, self.ny, self.nx, self.channels_last)]

    def forward_batch(self, voxel_features, coors, batch_size):
        return pillars_scatter(voxel_features, coors, batch_size, self.ny, self.nx, self.channels_last)
