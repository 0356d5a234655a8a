"""``bev_pool`` op -- host-side mirror of ``mmdet3d/ops/bev_pool/bev_pool.py``.

Same names, arguments and autograd behaviour as the reference
(``bev_pool(feats, coords, B, D, H, W) -> f32[B, C, D, H, W]``,
``QuickCumsumCuda.apply(x, geom_feats, ranks, B, D, H, W)``; only ``feats`` is
differentiable, bev_pool.py:81), but the extension calls go to the gfx950 C ABI
(``dbev_bev_pool_forward/backward``, include/dbev_hip.h) on torch's current stream.
"""
import torch

from . import _lib as L

__all__ = ["bev_pool", "QuickCumsumCuda", "bev_pool_forward", "bev_pool_backward"]


def bev_pool_forward(x, geom_feats, interval_lengths, interval_starts, B, D, H, W):
    """ext ``bev_pool_forward`` (bev_pool.cpp:22-47): -> f32[B, D, H, W, C]."""
    dev = L.require_cuda(x, geom_feats, interval_lengths, interval_starts)
    x = x.contiguous()
    assert x.dtype == torch.float32 and geom_feats.dtype == torch.int32
    assert interval_lengths.dtype == torch.int32 and interval_starts.dtype == torch.int32
    geom_feats = geom_feats.contiguous()
    interval_starts = interval_starts.contiguous()
    interval_lengths = interval_lengths.contiguous()
    n, c = x.shape
    out = torch.empty((B, D, H, W, c), dtype=x.dtype, device=dev)
    with torch.cuda.device(dev):
        L.call(
            "dbev_bev_pool_forward", L.ptr(x), L.ptr(geom_feats), L.ptr(interval_starts), L.ptr(interval_lengths),
            L.ptr(out), n, c, interval_starts.numel(),
            int(B), int(D), int(H), int(W), L.stream_ptr(dev))
    return out


def bev_pool_backward(out_grad, geom_feats, interval_lengths, interval_starts, B, D, H, W):
    """ext ``bev_pool_backward`` (bev_pool.cpp:60-87): -> f32[n, C]."""
    dev = L.require_cuda(out_grad, geom_feats)
    out_grad = out_grad.contiguous()
    n = geom_feats.shape[0]
    c = out_grad.shape[-1]
    x_grad = torch.empty((n, c), dtype=out_grad.dtype, device=dev)
    with torch.cuda.device(dev):
        L.call(
            "dbev_bev_pool_backward", L.ptr(out_grad), L.ptr(geom_feats), L.ptr(interval_starts), L.ptr(interval_lengths),
            L.ptr(x_grad), n, c, interval_starts.numel(), int(B), int(D), int(H), int(W),
            L.stream_ptr(dev))
    return x_grad


class QuickCumsumCuda(torch.autograd.Function):
    """bev_pool.py:37-81 (name kept from the reference; nothing CUDA about it here)."""

    @staticmethod
    def forward(ctx, x, geom_feats, ranks, B, D, H, W):
        kept = torch.ones(x.shape[0], device=x.device, dtype=torch.bool)
        kept[1:] = ranks[1:] != ranks[:-1]
        interval_starts = torch.where(kept)[0].int()
        interval_lengths = torch.zeros_like(interval_starts)
        interval_lengths[:-1] = interval_starts[1:] - interval_starts[:-1]
        if interval_starts.numel() > 0:
            interval_lengths[-1] = x.shape[0] - interval_starts[-1]
        geom_feats = geom_feats.int().contiguous()
        out = bev_pool_forward(x, geom_feats, interval_lengths, interval_starts, B, D, H, W)
        ctx.save_for_backward(interval_starts, interval_lengths, geom_feats)
        ctx.saved_shapes = B, D, H, W
        return out

    @staticmethod
    def backward(ctx, out_grad):
        interval_starts, interval_lengths, geom_feats = ctx.saved_tensors
        B, D, H, W = ctx.saved_shapes
        x_grad = bev_pool_backward(out_grad.contiguous(), geom_feats, interval_lengths,
                                   interval_starts, B, D, H, W)
        return x_grad, None, None, None, None, None, None


def bev_pool(feats, coords, B, D, H, W):
    """bev_pool.py:83-97.  feats f32[n, C]; coords int[n, 4] = (x, y, z, b) with
    0<=x<H, 0<=y<W, 0<=z<D -> f32[B, C, D, H, W]."""
    assert feats.shape[0] == coords.shape[0]
    B, D, H, W = int(B), int(D), int(H), int(W)
    ranks = (coords[:, 0] * (W * D * B) + coords[:, 1] * (D * B)
             + coords[:, 2] * B + coords[:, 3])
    # stable -> run-to-run deterministic summation order (the reference's argsort is not)
    indices = ranks.argsort(stable=True)
    feats, coords, ranks = feats[indices], coords[indices], ranks[indices]
    x = QuickCumsumCuda.apply(feats, coords, ranks, B, D, H, W)
    x = x.permute(0, 4, 1, 2, 3).contiguous()
    return x
