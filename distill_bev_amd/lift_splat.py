"""Fused Lift-Splat op (host side).

Python call surface for ``dbev_lift_splat_*`` / ``dbev_splat_*`` (include/dbev_hip.h):

    prep = lift_splat_prepare(geom, dx, bx, nx)       # geometry -> voxel index + CSR (no grad)
    bev  = lift_splat(depth_prob, img_feat, prep)     # == voxel_pooling(geom, depth x feat)
    bev  = voxel_pooling(volume, prep)                # == vt_mine.voxel_pooling(geom, volume)

``bev`` has the reference's logical shape f32[B, C*Z, Y, X]
(view_transformer_mine.py:176-178) with channels-last strides (the kernels write the
cell-major image directly; downstream convolutions consume it as is).
"""
import torch
from torch.autograd import Function

from . import _lib as L
from . import lss as LSS


class LiftSplatPrep:
    """Per-geometry state shared by forward and backward (and by the two BEVDepth4D frames'
    own calls): voxel index per frustum point and the cell -> points CSR."""
    __slots__ = ("point_cell", "cell_start", "cell_points", "n_kept", "hot_cells", "n_hot", "B", "X", "Y",
                 "Z", "n_points", "n_cells")

    def voxel_indices(self):
        """Decoded (x, y, z, b) int32[n_points, 4] with -1 rows for dropped points --
        the reference's ``geom_feats`` before the `kept` filter (vt_mine.py:150-160)."""
        c = self.point_cell.long()
        ok = c >= 0
        z = c % self.Z
        x = (c // self.Z) % self.X
        y = (c // (self.Z * self.X)) % self.Y
        b = c // (self.Z * self.X * self.Y)
        out = torch.stack([x, y, z, b], 1)
        out[~ok] = -1
        return out.int()


def lift_splat_prepare(geom, dx, bx, nx):
    """geom f32[B, N, D, H, W, 3] (device) ; dx, bx: 3 python floats / CPU tensors as produced by
    gen_dx_bx ; nx: 3 ints (X, Y, Z)."""
    dev = L.require_cuda(geom)
    geom = geom.contiguous()
    assert geom.dtype == torch.float32 and geom.shape[-1] == 3
    B = geom.shape[0]
    n_points = geom.numel() // 3
    X, Y, Z = (int(v) for v in nx)
    n_cells = B * X * Y * Z
    p = LiftSplatPrep()
    p.B, p.X, p.Y, p.Z, p.n_points, p.n_cells = B, X, Y, Z, n_points, n_cells
    p.point_cell = torch.empty((n_points,), dtype=torch.int32, device=dev)
    p.cell_start = torch.empty((n_cells + 1,), dtype=torch.int32, device=dev)
    p.cell_points = torch.empty((max(n_points, 1),), dtype=torch.int32, device=dev)
    p.n_kept = torch.empty((1,), dtype=torch.int32, device=dev)
    p.hot_cells = torch.empty((n_cells,), dtype=torch.int32, device=dev)
    p.n_hot = torch.empty((1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        nbytes = L.call("dbev_lift_splat_workspace_bytes", n_points, n_cells)
        ws = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=dev)
        L.call("dbev_lift_splat_prepare", L.ptr(geom), n_points, B,
               L.host_floats([float(v) for v in dx]), L.host_floats([float(v) for v in bx]),
               L.host_ints([X, Y, Z]), L.ptr(p.point_cell), L.ptr(p.cell_start), L.ptr(p.cell_points),
               L.ptr(p.n_kept), L.ptr(p.hot_cells), L.ptr(p.n_hot), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    return p


def camera_params(rots, trans, intrins, post_rots, post_trans):
    """f32[B*N, 24] per camera: inverse(post_rots) (9), post_trans (3), rots @ inverse(intrins) (9), trans (3)
    -- the matrices of get_geometry (vt_mine.py:121-136), a few hundred bytes of torch work per step."""
    B, N = trans.shape[:2]
    a = LSS.inverse_nosync(post_rots).reshape(B * N, 9)
    c = rots.matmul(LSS.inverse_nosync(intrins)).reshape(B * N, 9)
    return torch.cat([a, post_trans.reshape(B * N, 3), c, trans.reshape(B * N, 3)], dim=1).contiguous()


def lift_splat_prepare_cam(frustum, rots, trans, intrins, post_rots, post_trans, dx, bx, nx, cam=None):
    """lift_splat_prepare with get_geometry fused into the index kernel (no [B,N,D,H,W,3] tensor).
    cam: optional precomputed camera_params(...) tensor f32[B*N, 24] (e.g. a static buffer when the call is captured
    into a HIP graph: the batched 3x3 inverses of camera_params go through rocSOLVER, which is not capturable)."""
    dev = L.require_cuda(frustum, rots)
    B, N = trans.shape[:2]
    D, H, W = frustum.shape[:3]
    if cam is None:
        cam = camera_params(rots, trans, intrins, post_rots, post_trans)
    assert cam.shape == (B * N, 24) and cam.is_contiguous()
    fr = frustum.contiguous()
    n_points = B * N * D * H * W
    X, Y, Z = (int(v) for v in nx)
    n_cells = B * X * Y * Z
    p = LiftSplatPrep()
    p.B, p.X, p.Y, p.Z, p.n_points, p.n_cells = B, X, Y, Z, n_points, n_cells
    p.point_cell = torch.empty((n_points,), dtype=torch.int32, device=dev)
    p.cell_start = torch.empty((n_cells + 1,), dtype=torch.int32, device=dev)
    p.cell_points = torch.empty((max(n_points, 1),), dtype=torch.int32, device=dev)
    p.n_kept = torch.empty((1,), dtype=torch.int32, device=dev)
    p.hot_cells = torch.empty((n_cells,), dtype=torch.int32, device=dev)
    p.n_hot = torch.empty((1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        nbytes = L.call("dbev_lift_splat_workspace_bytes", n_points, n_cells)
        ws = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=dev)
        L.call("dbev_lift_splat_prepare_cam", L.ptr(cam), L.ptr(fr), B * N, D, H, W, B,
               L.host_floats([float(v) for v in dx]), L.host_floats([float(v) for v in bx]),
               L.host_ints([X, Y, Z]), L.ptr(p.point_cell), L.ptr(p.cell_start), L.ptr(p.cell_points),
               L.ptr(p.n_kept), L.ptr(p.hot_cells), L.ptr(p.n_hot), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    return p


def _as_logical(out, prep, C):
    # physical [B, Y, X, Z, C] -> logical [B, Z*C, Y, X] with channels-last strides
    return out.view(prep.B, prep.Y, prep.X, prep.Z * C).permute(0, 3, 1, 2)


def _as_physical(grad, prep, C):
    return grad.permute(0, 2, 3, 1).contiguous().view(prep.n_cells, C)


class _LiftSplat(Function):
    @staticmethod
    def forward(ctx, depth, feat, prep):
        dev = L.require_cuda(depth, feat)
        depth = depth.contiguous()
        BN, D, H, W = depth.shape
        C = feat.shape[1]
        assert feat.shape == (BN, C, H, W) and BN * D * H * W == prep.n_points
        feat_cl = feat.contiguous(memory_format=torch.channels_last)   # physical [BN, H, W, C]
        out = torch.empty((prep.n_cells, C), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.call("dbev_lift_splat_forward", L.ptr(depth), L.ptr(feat_cl), L.ptr(prep.cell_start),
                   L.ptr(prep.cell_points), L.ptr(prep.hot_cells), L.ptr(prep.n_hot), L.ptr(out), BN, D, H, W, C,
                   prep.n_cells, L.stream_ptr(dev))
        ctx.save_for_backward(depth, feat_cl)
        ctx.prep = prep
        return _as_logical(out, prep, C)

    @staticmethod
    def backward(ctx, grad_bev):
        depth, feat_cl = ctx.saved_tensors
        prep = ctx.prep
        BN, D, H, W = depth.shape
        C = feat_cl.shape[1]
        dev = depth.device
        g = _as_physical(grad_bev, prep, C)
        grad_depth = torch.empty_like(depth)
        grad_feat = torch.empty((BN, C, H, W), dtype=torch.float32, device=dev,
                                memory_format=torch.channels_last)
        with torch.cuda.device(dev):
            L.call("dbev_lift_splat_backward", L.ptr(g), L.ptr(depth), L.ptr(feat_cl),
                   L.ptr(prep.point_cell), L.ptr(grad_depth), L.ptr(grad_feat), BN, D, H, W, C,
                   L.stream_ptr(dev))
        return grad_depth, grad_feat, None


def lift_splat(depth, feat, prep):
    """depth f32[B*N, D, H, W] (softmaxed), feat f32[B*N, C, H, W] -> f32[B, C*Z, Y, X]."""
    return _LiftSplat.apply(depth, feat, prep)


class _VoxelPooling(Function):
    @staticmethod
    def forward(ctx, x, prep):
        dev = L.require_cuda(x)
        C = x.shape[-1]
        xf = x.reshape(-1, C).contiguous()
        assert xf.shape[0] == prep.n_points
        out = torch.empty((prep.n_cells, C), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.call("dbev_splat_forward", L.ptr(xf), L.ptr(prep.cell_start), L.ptr(prep.cell_points),
                   L.ptr(prep.hot_cells), L.ptr(prep.n_hot), L.ptr(out), prep.n_points, C, prep.n_cells,
                   L.stream_ptr(dev))
        ctx.prep = prep
        ctx.xshape = tuple(x.shape)
        return _as_logical(out, prep, C)

    @staticmethod
    def backward(ctx, grad_bev):
        prep = ctx.prep
        C = ctx.xshape[-1]
        dev = grad_bev.device
        g = _as_physical(grad_bev, prep, C)
        gx = torch.empty((prep.n_points, C), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.call("dbev_splat_backward", L.ptr(g), L.ptr(prep.point_cell), L.ptr(gx), prep.n_points, C,
                   L.stream_ptr(dev))
        return gx.view(ctx.xshape), None


def voxel_pooling(x, prep):
    """x f32[B, N, D, H, W, C] volume -> f32[B, C*Z, Y, X] (vt_mine.voxel_pooling semantics)."""
    return _VoxelPooling.apply(x, prep)
