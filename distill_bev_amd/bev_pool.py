"""``bev_pool`` op -- host-side mirror of ``mmdet3d/ops/bev_pool/bev_pool.py``.

Same names, arguments and autograd behaviour as the reference
(``bev_pool(feats, coords, B, D, H, W) -> f32[B, C, D, H, W]``,
``QuickCumsumCuda.apply(x, geom_feats, ranks, B, D, H, W)``; only ``feats`` is
differentiable, bev_pool.py:81), but the extension calls go to the gfx950 C ABI
(``dbev_bev_pool_forward/backward``, include/dbev_hip.h) on torch's current stream.
"""
import torch

from . import _lib as L

__all__ = ["bev_pool", "bev_pool_sorted", "QuickCumsumCuda", "bev_pool_forward", "bev_pool_backward"]


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


class _BevPoolCSR(torch.autograd.Function):
    """bev_pool without the argsort + row gather: cell -> point-list CSR from the integer coordinates
    (dbev_bev_pool_prepare), rows summed in place by dbev_splat_forward, gradient rows written by
    dbev_splat_backward.  Summation order inside a cell = ascending point index (deterministic)."""

    @staticmethod
    def forward(ctx, feats, coords, B, D, H, W):
        dev = L.require_cuda(feats, coords)
        feats = feats.contiguous()
        n, C = feats.shape
        n_cells = B * D * H * W
        i64 = coords.dtype == torch.int64
        coords = coords.contiguous() if i64 else coords.to(torch.int32).contiguous()    # int64 is consumed as it is
        point_cell = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
        cell_start = torch.empty((n_cells + 1,), dtype=torch.int32, device=dev)
        cell_points = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
        n_kept = torch.empty((1,), dtype=torch.int32, device=dev)
        hot_cells = torch.empty((n_cells,), dtype=torch.int32, device=dev)
        n_hot = torch.empty((1,), dtype=torch.int32, device=dev)
        out = torch.empty((B, D, H, W, C), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nbytes = L.call("dbev_lift_splat_workspace_bytes", n, n_cells)
            ws = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=dev)
            L.call("dbev_bev_pool_prepare_i64" if i64 else "dbev_bev_pool_prepare", L.ptr(coords), n, B, D, H, W,
                   L.ptr(point_cell), L.ptr(cell_start),
                   L.ptr(cell_points), L.ptr(n_kept), L.ptr(hot_cells), L.ptr(n_hot), L.ptr(ws), ws.numel(),
                   L.stream_ptr(dev))
            L.call("dbev_splat_forward", L.ptr(feats), L.ptr(cell_start), L.ptr(cell_points), L.ptr(hot_cells),
                   L.ptr(n_hot), L.ptr(out), n, C, n_cells, L.stream_ptr(dev))
        ctx.save_for_backward(point_cell)
        ctx.dims = (n, C)
        # the reference returns x.permute(0, 4, 1, 2, 3).contiguous(); here the SAME logical [B, C, D, H, W] tensor is a
        # zero-copy view of the cell-major rows (channels-last-3d strides) -- one 67 MB read + write less per call
        return out.permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, out_grad):
        (point_cell,) = ctx.saved_tensors
        n, C = ctx.dims
        dev = out_grad.device
        B, _, D, H, W = out_grad.shape
        g = out_grad.permute(0, 2, 3, 4, 1)            # the kernels read cell-major rows [B, D, H, W, C]
        if not g.is_contiguous():
            if out_grad.is_contiguous() and out_grad.dtype == torch.float32:      # the reference's [B, C, D, H, W] layout
                gt = torch.empty((B, D, H, W, C), dtype=torch.float32, device=dev)
                with torch.cuda.device(dev):
                    L.call("dbev_transpose_bcs_to_bsc", L.ptr(out_grad), L.ptr(gt), B, C, D * H * W, L.stream_ptr(dev))
                g = gt
            else:
                g = g.contiguous()
        out_grad = g
        x_grad = torch.empty((n, C), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.call("dbev_splat_backward", L.ptr(out_grad), L.ptr(point_cell), L.ptr(x_grad), n, C, L.stream_ptr(dev))
        return x_grad, None, None, None, None, None


def bev_pool(feats, coords, B, D, H, W, contiguous=False):
    """bev_pool.py:83-97.  feats f32[n, C]; coords int[n, 4] = (x, y, z, b) with
    0<=x<H, 0<=y<W, 0<=z<D -> f32[B, C, D, H, W].

    Layout contract: the LOGICAL shape is always the reference's [B, C, D, H, W]; the strides are channels-last-3d (the
    cell-major rows [B, D, H, W, C] the kernels write, viewed without a copy) on BOTH paths below, so whether a caller's
    `.view()` works never depends on the channel count.  `contiguous=True` returns the reference's physical layout
    (`x.permute(0, 4, 1, 2, 3).contiguous()`, bev_pool.py:95) at the price of one transposing copy.

    The reference ranks the points, argsorts, gathers `feats[indices]` and sums rank intervals
    (QuickCumsumCuda, kept above for callers of that level).  Same result here without moving the feature rows:
    see _BevPoolCSR (C must be a multiple of 4 and <= 256 for the row kernels; other widths take the
    sorted-interval path)."""
    assert feats.shape[0] == coords.shape[0]
    B, D, H, W = int(B), int(D), int(H), int(W)
    C = feats.shape[1]
    if feats.dtype == torch.float32 and C % 4 == 0 and C <= 256 and feats.shape[0] > 0:
        out = _BevPoolCSR.apply(feats, coords, B, D, H, W)
    else:
        out = bev_pool_sorted(feats, coords, B, D, H, W, contiguous=False)
    return out.contiguous() if contiguous else out


def bev_pool_sorted(feats, coords, B, D, H, W, contiguous=True):
    """the reference's own sequence (rank, argsort, gather, interval sums) on the HIP interval kernels"""
    B, D, H, W = int(B), int(D), int(H), int(W)
    ranks = (coords[:, 0] * (W * D * B) + coords[:, 1] * (D * B)
             + coords[:, 2] * B + coords[:, 3])
    # stable -> run-to-run deterministic summation order (the reference's argsort is not)
    indices = ranks.argsort(stable=True)
    feats, coords, ranks = feats[indices], coords[indices], ranks[indices]
    x = QuickCumsumCuda.apply(feats, coords, ranks, B, D, H, W)        # cell-major rows [B, D, H, W, C]
    x = x.permute(0, 4, 1, 2, 3)
    return x.contiguous() if contiguous else x
