"""Foreground-masked BEV feature distillation (FGD) -- host side.

Mirrors the pieces of ``mmdet3d/models/detectors/bevdet_distill.py`` that sit on the hot path
(``foreground_scale_mask`` :755-843, the attention / mask algebra and the three masked-MSE
sums of ``fgd_distill_loss`` :1084-1108,1110-1129,1163-1168,1253-1262,1282-1287) on top of the
gfx950 kernels ``dbev_fg_scale_mask``, ``dbev_abs_mean_maps``, ``dbev_fgd_masked_mse_*``.

The O(M) box -> face-plane setup stays on the host in float32 numpy with the exact operation
order of ``mmdet3d/core/bbox/box_np_ops.py`` (the GT boxes are host data in the reference
too: ``DataContainer(cpu_only=True)``); the O(H*W*M) rasterisation and every pass over the
feature maps run on the GPU.  No mask ever travels host<->device.
"""
import numpy as np
import torch
from torch.autograd import Function

from . import _lib as L

_SURF = np.array([[0, 1, 2, 3], [7, 6, 5, 4], [0, 3, 7, 4], [1, 5, 6, 2], [0, 4, 5, 1], [3, 2, 6, 7]])
_PATTERN = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 1], [0, 1, 0],
                     [1, 0, 0], [1, 0, 1], [1, 1, 1], [1, 1, 0]], dtype=np.float32)


def box_face_planes(boxes):
    """boxes f32[M, 7] (x, y, z_bottom, w, l, h, yaw) -> f32[M, 6, 4] = (nx, ny, nz, d) with
    n.p + d < 0 for interior points.  float32 numpy, same steps as box_np_ops
    center_to_corner_box3d(origin=(0.5,0.5,0), axis=2) :206-235, corner_to_surfaces_3d
    :404-423 and surface_equ_3d :694-715."""
    b = np.ascontiguousarray(boxes[:, :7], dtype=np.float32)
    pat = _PATTERN - np.array([0.5, 0.5, 0.0], dtype=np.float32)
    corners = b[:, None, 3:6] * pat[None]
    s, c = np.sin(b[:, 6]), np.cos(b[:, 6])
    z, o = np.zeros_like(c), np.ones_like(c)
    rot_t = np.stack([[c, -s, z], [s, c, z], [z, z, o]])
    corners = np.einsum("aij,jka->aik", corners, rot_t) + b[:, None, :3]
    surf = corners[:, _SURF]
    vec = surf[:, :, :2] - surf[:, :, 1:3]
    normal = np.cross(vec[:, :, 0], vec[:, :, 1])
    d = -np.einsum("aij,aij->ai", normal, surf[:, :, 0])
    return np.concatenate([normal, d[..., None]], -1).astype(np.float32)


class ForegroundMaskRasterizer:
    """``foreground_scale_mask`` bound to one head train_cfg (grid_size / point_cloud_range /
    voxel_size, bevdet_distill.py:756-758).  Caches the per-resolution cell coordinates."""

    def __init__(self, grid_size, point_cloud_range, voxel_size, cell_center=False):
        # cell_center: the BEVFormer variant (bevformer_distill.py:409-419) -- a fractional out_size_factor
        # (grid 512 over 200 cells) and coordinates at the cell centres instead of the lower cell corners
        self.cell_center = cell_center
        self.grid_size = torch.tensor(grid_size)
        self.pc_range = torch.tensor(point_cloud_range, dtype=torch.float32)
        self.voxel_size = torch.tensor(voxel_size, dtype=torch.float32)
        self._coords = {}

    def _cell_coords(self, H, W, dev):
        key = (H, W, str(dev))
        if key not in self._coords:
            assert int(self.grid_size[0]) == int(self.grid_size[1]) and H == W
            if self.cell_center:
                osf = self.grid_size[0] / W
                xs = torch.stack([i * self.voxel_size[0] * osf + self.pc_range[0] + self.voxel_size[0] * osf / 2 for i in range(W)])
                ys = torch.stack([i * self.voxel_size[1] * osf + self.pc_range[1] + self.voxel_size[1] * osf / 2 for i in range(H)])
            else:
                assert int(self.grid_size[0]) % W == 0
                osf = self.grid_size[0] // W
                # same 0-dim float32 tensor arithmetic as bevdet_distill.py:766-767
                xs = torch.stack([i * self.voxel_size[0] * osf + self.pc_range[0] for i in range(W)])
                ys = torch.stack([i * self.voxel_size[1] * osf + self.pc_range[1] for i in range(H)])
            area = self.voxel_size[0] * self.voxel_size[1] * osf * osf
            self._coords[key] = (xs.float().to(dev), ys.float().to(dev), area)
        return self._coords[key]

    def __call__(self, H, W, gt_boxes, device):
        """gt_boxes: list (one per sample) of f32[M_b, >=7] arrays / CPU tensors in the
        LiDARInstance3DBoxes.tensor layout.  -> fg, fg_scale, bg_scale f32[B,1,H,W] on device."""
        xs, ys, area = self._cell_coords(H, W, device)
        planes, scales, offs = [], [], [0]
        for boxes in gt_boxes:
            b = np.array(boxes.numpy() if torch.is_tensor(boxes) else boxes, dtype=np.float32)[:, :7].copy()
            b[:, 2] = 0      # :785-786 unify z: bottom 0, height 1
            b[:, 5] = 1
            planes.append(box_face_planes(b).reshape(-1, 4))
            # sqrt(area / (w * l)) in float32 (:805-806).  The reference calls torch.sqrt on the
            # CPU, whose vectorised kernel is NOT correctly rounded and differs by 1 ulp between
            # hosts; the correctly rounded numpy sqrt is used here (<= 1 ulp from any of them).
            scales.append(np.sqrt((np.float32(area.item()) / (b[:, 3] * b[:, 4])).astype(np.float32)))
            offs.append(offs[-1] + b.shape[0])
        B = len(gt_boxes)
        pl = np.concatenate(planes) if offs[-1] else np.zeros((1, 4), np.float32)
        sc = np.concatenate(scales) if offs[-1] else np.zeros((1,), np.float32)
        # the sources are pageable temporaries: staged through pinned memory (L.h2d), not copied straight (a pageable H2D blocks
        # the host until the GPU queue has drained)
        pl_d = L.h2d(torch.from_numpy(np.ascontiguousarray(pl)), device)
        sc_d = L.h2d(torch.from_numpy(np.ascontiguousarray(sc, dtype=np.float32)), device)
        of_d = L.h2d(torch.tensor(offs, dtype=torch.int32), device)
        fg = torch.empty((B, 1, H, W), dtype=torch.float32, device=device)
        fs = torch.empty_like(fg)
        bs = torch.empty_like(fg)
        cnt = torch.empty((B,), dtype=torch.int32, device=device)
        with torch.cuda.device(device):
            L.call("dbev_fg_scale_mask", L.ptr(pl_d), L.ptr(sc_d), L.ptr(of_d), L.ptr(xs), L.ptr(ys),
                   B, H, W, L.ptr(fg), L.ptr(fs), L.ptr(bs), L.ptr(cnt), L.stream_ptr(device))
        return fg, fs, bs


def _is_nhwc(t):
    """physically [B, H, W, C] (and not also NCHW-contiguous), fp32, channel count the NHWC kernels take"""
    return (t.dim() == 4 and t.dtype == torch.float32 and t.is_contiguous(memory_format=torch.channels_last)
            and not t.is_contiguous() and t.shape[1] % 4 == 0 and t.shape[1] <= 1024)


def abs_mean_maps(x, with_pool=False):
    """x f32[B, C, H, W] -> (mean_c |x| f32[B,1,H,W], mean_hw |x| f32[B,C,1,1]) in one pass.
    with_pool=True: also mean_c x f32[B,1,H,W] (no autograd; None when the layout has no fused path)."""
    dev = L.require_cuda(x)
    B, C, H, W = x.shape
    pix = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
    ch = torch.empty((B, C, 1, 1), dtype=torch.float32, device=dev)
    if _is_nhwc(x):                       # channels-last activations are consumed as they are
        pool = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev) if with_pool else None
        with torch.cuda.device(dev):
            nbytes = L.call("dbev_abs_mean_maps_nhwc_workspace_bytes", B, C, H * W)
            ws = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=dev)
            L.call("dbev_abs_mean_maps_nhwc", L.ptr(x), B, C, H * W, L.ptr(pix), L.ptr(ch), L.ptr(pool), L.ptr(ws),
                   ws.numel(), L.stream_ptr(dev))
        return (pix, ch, pool) if with_pool else (pix, ch)
    x = x.contiguous()
    with torch.cuda.device(dev):
        nbytes = L.call("dbev_abs_mean_maps_workspace_bytes", B, C, H * W)
        ws = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=dev)
        L.call("dbev_abs_mean_maps", L.ptr(x), B, C, H * W, L.ptr(pix), L.ptr(ch), L.ptr(ws), ws.numel(),
               L.stream_ptr(dev))
    return (pix, ch, None) if with_pool else (pix, ch)


class _MaskedMSE(Function):
    @staticmethod
    def forward(ctx, S, T, Wfg, Wbg, Wfp, Cc, s_pool=None):
        """s_pool (optional, NHWC path only): mean_c S precomputed by abs_mean_maps(S, with_pool=True); it is returned
        as a second, differentiable output whose gradient is folded into the dS kernel (+ grad / C per channel)."""
        dev = L.require_cuda(S, T, Wfg, Wbg)
        B, C, H, W = S.shape
        assert T.shape == S.shape
        Wfg = Wfg.contiguous(); Wbg = Wbg.contiguous()
        Wfp = Wfp.contiguous() if Wfp is not None else None
        Cc = Cc.contiguous() if Cc is not None else None
        out = torch.empty((3,), dtype=torch.float32, device=dev)
        ctx.nhwc = _is_nhwc(S)
        if ctx.nhwc:
            T = T.contiguous(memory_format=torch.channels_last)
            with torch.cuda.device(dev):
                nbytes = L.call("dbev_fgd_masked_mse_nhwc_workspace_bytes", B, C, H * W)
                ws = torch.empty((max(int(nbytes), 16),), dtype=torch.uint8, device=dev)
                L.call("dbev_fgd_masked_mse_forward_nhwc", L.ptr(S), L.ptr(T), L.ptr(Wfg), L.ptr(Wbg), L.ptr(Wfp),
                       L.ptr(Cc), B, C, H * W, L.ptr(out), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
            ctx.save_for_backward(S, T, Wfg, Wbg, Wfp, Cc)
            ctx.pooled = s_pool is not None
            return (out, s_pool.view_as(s_pool)) if ctx.pooled else out
        assert s_pool is None, "the pooled-mean output exists on the channels-last path only"
        ctx.pooled = False
        S = S.contiguous()
        T = T.contiguous()
        with torch.cuda.device(dev):
            nbytes = L.call("dbev_fgd_masked_mse_workspace_bytes", B, C, H * W)
            if nbytes == 0:
                raise L.DbevHipError("fgd_masked_mse: H*W must be a positive multiple of 4")
            ws = torch.empty((int(nbytes),), dtype=torch.uint8, device=dev)
            L.call("dbev_fgd_masked_mse_forward", L.ptr(S), L.ptr(T), L.ptr(Wfg), L.ptr(Wbg), L.ptr(Wfp),
                   L.ptr(Cc), B, C, H * W, L.ptr(out), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
        ctx.save_for_backward(S, T, Wfg, Wbg, Wfp, Cc)
        return out

    @staticmethod
    def backward(ctx, grad_out, grad_pool=None):
        S, T, Wfg, Wbg, Wfp, Cc = ctx.saved_tensors
        B, C, H, W = S.shape
        dev = S.device
        g = (grad_out if grad_out is not None else torch.zeros((3,), device=dev)).contiguous().float()
        dS = torch.empty_like(S)          # keeps S's memory format
        with torch.cuda.device(dev):
            if ctx.nhwc:
                gp = grad_pool.contiguous().float() if (ctx.pooled and grad_pool is not None) else None
                L.call("dbev_fgd_masked_mse_backward_nhwc", L.ptr(S), L.ptr(T), L.ptr(Wfg), L.ptr(Wbg), L.ptr(Wfp),
                       L.ptr(Cc), L.ptr(g), L.ptr(gp), B, C, H * W, L.ptr(dS), L.stream_ptr(dev))
            else:
                L.call("dbev_fgd_masked_mse_backward", L.ptr(S), L.ptr(T), L.ptr(Wfg), L.ptr(Wbg), L.ptr(Wfp),
                       L.ptr(Cc), L.ptr(g), B, C, H * W, L.ptr(dS), L.stream_ptr(dev))
        return dS, None, None, None, None, None, None


def masked_mse_sums(S, T, Wfg, Wbg, Wfp=None, Cc=None, s_pool=None):
    """-> f32[3]: sum((S-T)^2 Wfg), sum((S-T)^2 Wbg), sum((S-T)^2 Wfp Cc) (0 if Wfp is None).
    Differentiable wrt S only (teacher and masks are detached in the reference).
    s_pool: see _MaskedMSE.forward -> returns (sums, differentiable mean_c S)."""
    return _MaskedMSE.apply(S, T, Wfg, Wbg, Wfp, Cc, s_pool)


def fgd_feature_losses(student_feat, teacher_feat, fg, fg_scale, bg_scale, *, w_fg, w_bg,
                       spatial_t=0.5, channel_t=0.5, s_ratio=1.0, spatial_att="teacher_student",
                       spatial_mask=True, channel_mask=False, fp=None, fp_scale=None, n_fp=None, w_fp=0.0):
    """kd_fg / kd_bg / kd_fp feature losses of fgd_distill_loss for scale_mask='combine_gt',
    foreground_mask='gt', background_mask='logical_not' (the recipe of
    scripts/teacher_to_bevdepth4d/centerpoint2bevdepth.sh:30-41).
    student_feat: adapted student features (requires grad); teacher_feat: no grad."""
    B, C, H, W = student_feat.shape
    teacher_feat = teacher_feat.detach()
    t_pix, t_ch, t_pool = abs_mean_maps(teacher_feat, with_pool=True)
    s_pool = None
    t_att = torch.softmax(t_pix.view(B, -1) / spatial_t, dim=1) * (H * W)
    if spatial_att == "teacher":
        att = t_att
    elif spatial_att == "teacher_student":
        s_pix, _, s_pool = abs_mean_maps(student_feat.detach(), with_pool=True)
        s_att = torch.softmax(s_pix.view(B, -1) / spatial_t, dim=1) * (H * W)
        att = (t_att + s_att * s_ratio) / (1 + s_ratio)
    else:
        raise NotImplementedError(spatial_att)
    att = att.view(B, 1, H, W).detach()
    c_att = (torch.softmax(t_ch.view(B, -1) / channel_t, dim=1) * C).view(B, C).detach()

    bg = (fg == 0).float()
    bgs = bg_scale
    if fp is not None:
        bg = bg * (fp == 0).float()
        n_bg = H * W - fg.sum(dim=(1, 2, 3))
        denom = n_bg - n_fp
        bgs = torch.where(denom > 0, 1.0 / denom.clamp(min=1), torch.zeros_like(denom)).view(B, 1, 1, 1)
        bgs = bgs.expand_as(bg_scale)
    scale = torch.maximum(fg_scale, bgs)
    w_f = fg * scale
    w_b = bg * scale
    if spatial_mask:
        w_f = w_f * att
        w_b = w_b * att
    cc = None
    if channel_mask:
        # per-channel factor on fg/bg too: fold into a 3-term call is not possible -> generic path
        raise NotImplementedError("channel_mask=True is not part of the hot-path recipe")
    w_p = None
    if fp is not None:
        w_p = (fp * fp_scale * att).contiguous()
        cc = c_att
    pools = None
    if t_pool is not None and s_pool is not None and _is_nhwc(student_feat):
        # channel means of both features came out of the attention pass; the student's re-enters autograd as a second
        # output of the masked-MSE op so that its gradient is added inside the same dS kernel
        sums, s_pool = masked_mse_sums(student_feat, teacher_feat, w_f.contiguous(), w_b.contiguous(), w_p, cc, s_pool)
        pools = (t_pool, s_pool)
    else:
        sums = masked_mse_sums(student_feat, teacher_feat, w_f.contiguous(), w_b.contiguous(), w_p, cc)
    out = {"kd_fg_feat_loss": sums[0] * (w_fg / B), "kd_bg_feat_loss": sums[1] * (w_bg / B)}
    if fp is not None:
        out["kd_fp_bg_feat_loss"] = sums[2] * (w_fp / B)
    return out, att, c_att, pools


class _FusedAdaptMSE(Function):
    """1x1-convolution adaptation + per-pixel reductions of the FGD loss on the fp32 matrix cores
    (dbev_adapt_mse_forward, csrc/adapt_mse.hip).  forward(x, weight, bias, teacher, cc) ->
      E    [B,1,H,W]  sum_c (s - t)^2            (differentiable)
      Efp  [B,1,H,W]  sum_c cc[b,c] (s - t)^2    (differentiable; == E when cc is None)
      A    [B,1,H,W]  mean_c |s|                 (attention input, no gradient -- detached in the reference)
      P    [B,1,H,W]  mean_c s                   (differentiable; spatial term)
    with s = conv1x1(x) never written to memory; the difference s - t is kept for the backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, teacher, cc):
        dev = L.require_cuda(x, weight, teacher)
        B, Cs, H, W = x.shape
        Ct = weight.shape[0]
        S = int(L.call("dbev_adapt_mse_map_slices", Cs, Ct))
        assert S > 0 and teacher.shape == (B, Ct, H, W)
        x = x.contiguous(memory_format=torch.channels_last)
        teacher = teacher.contiguous(memory_format=torch.channels_last)
        w2 = weight.reshape(Ct, Cs).contiguous()
        M = B * H * W
        D = torch.empty((B, Ct, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        maps = torch.empty((S, 4, M), dtype=torch.float32, device=dev)
        cc = cc.contiguous() if cc is not None else None
        with torch.cuda.device(dev):
            L.call("dbev_adapt_mse_forward", L.ptr(x), L.ptr(w2), L.ptr(bias.contiguous()), L.ptr(teacher), L.ptr(cc),
                   B, H * W, Cs, Ct, L.ptr(D), L.ptr(maps), L.stream_ptr(dev))
        m = maps.sum(0) if S > 1 else maps[0]
        E, Efp = m[0].view(B, 1, H, W), m[1].view(B, 1, H, W)
        A, P = (m[2] / Ct).view(B, 1, H, W), (m[3] / Ct).view(B, 1, H, W)
        ctx.save_for_backward(x, weight, D, cc)
        ctx.mark_non_differentiable(A)
        return E, Efp, A, P

    @staticmethod
    def backward(ctx, gE, gEfp, gA, gP):
        x, weight, D, cc = ctx.saved_tensors
        B, Ct, H, W = D.shape
        dev = D.device
        z = lambda g: g.contiguous().float() if g is not None else None
        gE = z(gE) if gE is not None else torch.zeros((B, 1, H, W), device=dev)
        dS = torch.empty_like(D)
        with torch.cuda.device(dev):
            L.call("dbev_adapt_mse_backward_ds", L.ptr(D), L.ptr(gE), L.ptr(z(gEfp)), L.ptr(z(gP)), L.ptr(cc), B, H * W, Ct,
                   L.ptr(dS), L.stream_ptr(dev))
        # the convolution gradients are plain GEMMs on dS: the bf16x6 kernels where they apply (gemm_bf6), MIOpen's fp32 kernels
        # otherwise; the bias gradient is a column sum of dS (colsum)
        from . import colsum, gemm_bf6 as G
        nx, nw, nb = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        nhwc = dS.is_contiguous(memory_format=torch.channels_last) and x.is_contiguous(memory_format=torch.channels_last)
        dx = G.data_gradient(dS, weight) if nx and nhwc else None
        dw = G.weight_gradient(x, dS, weight) if nw and nhwc else None
        db = colsum.channel_sum(dS) if nb and nhwc else None
        lx, lw, lb = nx and dx is None, nw and dw is None, nb and db is None
        if lx or lw or lb:
            a, b, c = torch.ops.aten.convolution_backward(dS, x, weight, [Ct], [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [lx, lw, lb])
            dx, dw, db = (a if lx else dx), (b if lw else dw), (c if lb else db)
        if dw is not None and dw.shape != weight.shape:
            dw = dw.view(weight.shape)
        return dx, dw, db, None, None


def fused_adapt_eligible(conv, student_in, teacher_feat):
    """the 'head' recipe: nn.Conv2d(Cs, Ct, 1) with bias on channels-last fp32 features, Cs % 32 == Ct % 32 == 0"""
    if type(conv).__name__ not in ("Conv2d", "BiasSumConv2d") or not isinstance(conv, torch.nn.Conv2d) or conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.padding != (0, 0) \
            or conv.groups != 1 or conv.bias is None or conv.dilation != (1, 1):
        return False
    if not (student_in.is_cuda and student_in.dtype == torch.float32 and _is_nhwc(student_in) and _is_nhwc(teacher_feat)):
        return False
    return int(L.call("dbev_adapt_mse_map_slices", conv.in_channels, conv.out_channels)) > 0 \
        and teacher_feat.shape[1] == conv.out_channels and teacher_feat.shape[2:] == student_in.shape[2:]


def fgd_feature_losses_fused_adapt(student_in, conv, teacher_feat, fg, fg_scale, bg_scale, *, w_fg, w_bg, spatial_t=0.5,
                                   channel_t=0.5, s_ratio=1.0, spatial_att="teacher_student", spatial_mask=True,
                                   channel_mask=False, fp=None, fp_scale=None, n_fp=None, w_fp=0.0):
    """fgd_feature_losses with the student adaptation folded in (see _FusedAdaptMSE): same recipe, same outputs; the
    adapted student tensor does not exist, the three masked sums are per-pixel weighted sums of the kernel's maps."""
    if channel_mask:
        raise NotImplementedError("channel_mask=True is not part of the hot-path recipe")
    B, Ct, H, W = teacher_feat.shape
    teacher_feat = teacher_feat.detach()
    t_pix, t_ch, t_pool = abs_mean_maps(teacher_feat, with_pool=True)
    c_att = (torch.softmax(t_ch.view(B, -1) / channel_t, dim=1) * Ct).view(B, Ct).detach()
    E, Efp, A, P = _FusedAdaptMSE.apply(student_in, conv.weight, conv.bias, teacher_feat, c_att if fp is not None else None)
    t_att = torch.softmax(t_pix.view(B, -1) / spatial_t, dim=1) * (H * W)
    if spatial_att == "teacher":
        att = t_att
    elif spatial_att == "teacher_student":
        s_att = torch.softmax(A.view(B, -1) / spatial_t, dim=1) * (H * W)
        att = (t_att + s_att * s_ratio) / (1 + s_ratio)
    else:
        raise NotImplementedError(spatial_att)
    att = att.view(B, 1, H, W).detach()
    bg = (fg == 0).float()
    bgs = bg_scale
    if fp is not None:
        bg = bg * (fp == 0).float()
        n_bg = H * W - fg.sum(dim=(1, 2, 3))
        denom = n_bg - n_fp
        bgs = torch.where(denom > 0, 1.0 / denom.clamp(min=1), torch.zeros_like(denom)).view(B, 1, 1, 1).expand_as(bg_scale)
    scale = torch.maximum(fg_scale, bgs)
    w_f, w_b = fg * scale, bg * scale
    if spatial_mask:
        w_f, w_b = w_f * att, w_b * att
    out = {"kd_fg_feat_loss": (E * w_f).sum() * (w_fg / B), "kd_bg_feat_loss": (E * w_b).sum() * (w_bg / B)}
    if fp is not None:
        out["kd_fp_bg_feat_loss"] = (Efp * (fp * fp_scale * att)).sum() * (w_fp / B)
    return out, att, c_att, (t_pool, P)


class _UpsampleBilinearAC(Function):
    @staticmethod
    def forward(ctx, x, scale):
        dev = L.require_cuda(x)
        B, C, IH, IW = x.shape
        OH, OW = int(IH * scale), int(IW * scale)
        cl = x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous() and C % 4 == 0
        x = x.contiguous(memory_format=torch.channels_last) if cl else x.contiguous()
        y = torch.empty((B, C, OH, OW), dtype=torch.float32, device=dev,
                        memory_format=torch.channels_last if cl else torch.contiguous_format)
        with torch.cuda.device(dev):
            L.call("dbev_upsample_bilinear_ac_forward", L.ptr(x), L.ptr(y), B, C, IH, IW, OH, OW, 1 if cl else 0,
                   L.stream_ptr(dev))
        ctx.dims = (B, C, IH, IW, OH, OW, cl)
        return y

    @staticmethod
    def backward(ctx, gy):
        B, C, IH, IW, OH, OW, cl = ctx.dims
        dev = gy.device
        gy = gy.contiguous(memory_format=torch.channels_last) if cl else gy.contiguous()
        gx = torch.empty((B, C, IH, IW), dtype=torch.float32, device=dev,
                         memory_format=torch.channels_last if cl else torch.contiguous_format)
        with torch.cuda.device(dev):
            L.call("dbev_upsample_bilinear_ac_backward", L.ptr(gy), L.ptr(gx), B, C, IH, IW, OH, OW, 1 if cl else 0,
                   L.stream_ptr(dev))
        return gx, None


class UpsampleBilinearAC(torch.nn.Module):
    """nn.Upsample(scale_factor=s, mode='bilinear', align_corners=True) on the gfx950 kernel
    (parameter-free, so checkpoints are unaffected)."""

    def __init__(self, scale_factor):
        super().__init__()
        self.scale_factor = scale_factor

    def forward(self, x):
        return _UpsampleBilinearAC.apply(x, self.scale_factor)

    def extra_repr(self):
        return f"scale_factor={self.scale_factor}, mode=bilinear, align_corners=True"
