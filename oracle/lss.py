"""Oracle (CPU, numpy) for the Lift-Splat-Shoot view transform.  TEST INFRASTRUCTURE.

Restates, function by function, the reference's
``mmdet3d/models/necks/view_transformer_mine.py`` (vt_mine) and
``mmdet3d/ops/bev_pool`` (bev_pool.py + src/bev_pool_cuda.cu).  All arithmetic
that decides an *index* is done in float32 exactly in the order torch does it;
all *sums* are offered both as an fp64-exact segment sum (the yardstick) and
as the reference's own order of operations where that is well defined.

Pinned by tests/golden/lss_*.npz (generated from the imported reference).
"""
import numpy as np

f32 = np.float32


# --------------------------------------------------------------------------
# grid constants: vt_mine.py:14-18 gen_dx_bx
# --------------------------------------------------------------------------
def gen_dx_bx(xbound, ybound, zbound):
    """vt_mine.py:14-18.  Python-float arithmetic, then a cast to fp32
    (``torch.Tensor([...])``)."""
    rows = [xbound, ybound, zbound]
    dx = np.array([r[2] for r in rows], dtype=f32)
    bx = np.array([r[0] + r[2] / 2.0 for r in rows], dtype=f32)
    nx = np.array([(r[1] - r[0]) / r[2] for r in rows], dtype=f32)
    return dx, bx, nx


def torch_linspace_f32(start, end, steps):
    """Bit-exact restatement of ``torch.linspace(start, end, steps,
    dtype=torch.float)`` on CPU (ATen RangeFactories: fp32 step, first half
    counted up from start, second half counted down from end; the vectorised
    kernel contracts ``start + step*i`` into an FMA = ONE rounding, which is
    reproduced here by forming the exact product/sum in fp64 and rounding once;
    verified against torch in tests/test_oracle_lss.py)."""
    start = f32(start)
    end = f32(end)
    out = np.empty(steps, dtype=f32)
    if steps == 1:
        out[0] = start
        return out
    step = f32((end - start) / f32(steps - 1))
    half = steps // 2
    for i in range(steps):
        if i < half:
            out[i] = f32(np.float64(start) + np.float64(step) * i)
        else:
            out[i] = f32(np.float64(end) - np.float64(step) * (steps - i - 1))
    return out


def create_frustum(input_size=(256, 704), downsample=16, dbound=(1.0, 60.0, 1.0)):
    """vt_mine.py:98-109.  Returns f32[D, fH, fW, 3] = (x_pix, y_pix, depth)."""
    ogfH, ogfW = input_size
    fH, fW = ogfH // downsample, ogfW // downsample
    ds = np.arange(dbound[0], dbound[1], dbound[2], dtype=np.float64).astype(f32)
    D = ds.shape[0]
    xs = torch_linspace_f32(0, ogfW - 1, fW)
    ys = torch_linspace_f32(0, ogfH - 1, fH)
    fr = np.empty((D, fH, fW, 3), dtype=f32)
    fr[..., 0] = xs[None, None, :]
    fr[..., 1] = ys[None, :, None]
    fr[..., 2] = ds[:, None, None]
    return fr


def _matvec3(M, p):
    """fp32 3x3 @ 3-vector over leading dims, products summed left to right
    (the tolerance tests do not rely on the last ulp of this)."""
    out = np.empty(p.shape, dtype=f32)
    for i in range(3):
        acc = (M[..., i, 0] * p[..., 0]).astype(f32)
        acc = (acc + (M[..., i, 1] * p[..., 1]).astype(f32)).astype(f32)
        acc = (acc + (M[..., i, 2] * p[..., 2]).astype(f32)).astype(f32)
        out[..., i] = acc
    return out


def get_geometry(frustum, rots, trans, intrins, post_rots, post_trans):
    """vt_mine.py:111-139.  rots/intrins/post_rots f32[B,N,3,3], trans/post_trans
    f32[B,N,3] -> f32[B,N,D,fH,fW,3] ego-frame points.

    Inverses are taken in fp64 and rounded to fp32 (torch uses an fp32 LU); the
    result therefore agrees with the reference to a few ulp, which is what the
    geometry test asserts (abs 1e-4 m).  Index bit-exactness is asserted
    separately on a *shared* geometry tensor (see voxel_index)."""
    B, N = trans.shape[:2]
    pts = frustum[None, None].astype(f32) - post_trans.reshape(B, N, 1, 1, 1, 3).astype(f32)
    inv_post = np.linalg.inv(post_rots.astype(np.float64)).astype(f32).reshape(B, N, 1, 1, 1, 3, 3)
    pts = _matvec3(inv_post, pts)
    pts = np.concatenate([(pts[..., :2] * pts[..., 2:3]).astype(f32), pts[..., 2:3]], axis=-1)
    inv_k = np.linalg.inv(intrins.astype(np.float64))
    combine = (rots.astype(np.float64) @ inv_k).astype(f32).reshape(B, N, 1, 1, 1, 3, 3)
    pts = _matvec3(combine, pts)
    pts = (pts + trans.reshape(B, N, 1, 1, 1, 3).astype(f32)).astype(f32)
    return pts


# --------------------------------------------------------------------------
# voxel index: vt_mine.py:150-160  (bit-exact, int)
# --------------------------------------------------------------------------
def voxel_index(geom, dx, bx, nx):
    """vt_mine.py:150: ``((geom - (bx - dx/2.)) / dx).long()`` -- fp32 subtract,
    fp32 IEEE divide, then truncation TOWARD ZERO (not floor).
    geom f32[...,3] -> (idx int64[...,3], kept bool[...]) with kept as
    vt_mine.py:157-159 (0 <= idx < nx on all three axes)."""
    geom = np.asarray(geom, dtype=f32)
    lo = (bx.astype(f32) - (dx.astype(f32) / f32(2.0)).astype(f32)).astype(f32)
    q = ((geom - lo).astype(f32) / dx.astype(f32)).astype(f32)
    idx = np.trunc(q).astype(np.int64)
    nxl = nx.astype(np.int64)
    kept = ((idx[..., 0] >= 0) & (idx[..., 0] < nxl[0]) &
            (idx[..., 1] >= 0) & (idx[..., 1] < nxl[1]) &
            (idx[..., 2] >= 0) & (idx[..., 2] < nxl[2]))
    return idx, kept


def ranks_vt(idx, batch_ix, nx, B):
    """vt_mine.py:164-167 rank used to group points of a voxel."""
    nxl = nx.astype(np.int64)
    return (idx[:, 0] * (nxl[1] * nxl[2] * B) + idx[:, 1] * (nxl[2] * B)
            + idx[:, 2] * B + batch_ix)


# --------------------------------------------------------------------------
# splat: vt_mine.py:141-181 voxel_pooling
# --------------------------------------------------------------------------
def voxel_pooling(geom, x, dx, bx, nx, exact=True):
    """vt_mine.py:141-181.  geom f32[B,N,D,H,W,3], x f32[B,N,D,H,W,C]
    -> f32[B, C*Z, Y, X].

    exact=True : per-voxel sum accumulated in fp64, rounded once to fp32 (the
                 yardstick; the reference's cumsum trick differs from this by
                 ~1.5e-5 at |out|<=2).
    exact=False: fp32 sequential sum in point-id order (the order a stable
                 sort gives; what the HIP kernels implement)."""
    B, N, D, H, W, C = x.shape
    Np = B * N * D * H * W
    xf = x.reshape(Np, C)
    idx, kept = voxel_index(geom.reshape(Np, 3), dx, bx, nx)
    batch_ix = np.repeat(np.arange(B, dtype=np.int64), Np // B)
    nxl = nx.astype(np.int64)
    X, Y, Z = int(nxl[0]), int(nxl[1]), int(nxl[2])
    acc_t = np.float64 if exact else f32
    final = np.zeros((B, Z, Y, X, C), dtype=acc_t)
    ii = idx[kept]
    bb = batch_ix[kept]
    lin = ((bb * Z + ii[:, 2]) * Y + ii[:, 1]) * X + ii[:, 0]
    flat = final.reshape(-1, C)
    if exact:
        np.add.at(flat, lin, xf[kept].astype(np.float64))
    else:
        xs = xf[kept]
        order = np.argsort(lin, kind="stable")
        lin_s = lin[order]
        xs = xs[order]
        # fp32 sequential sum per voxel in point-id order
        starts = np.flatnonzero(np.r_[True, lin_s[1:] != lin_s[:-1]])
        ends = np.r_[starts[1:], lin_s.shape[0]]
        for s, e in zip(starts, ends):
            acc = np.zeros(C, dtype=f32)
            for r in range(s, e):
                acc = (acc + xs[r]).astype(f32)
            flat[lin_s[s]] = acc
    final = final.astype(f32)
    # final[b, c, z, y, x] ; cat(unbind(z), 1) -> channel = z*C + c
    out = final.transpose(0, 1, 4, 2, 3).reshape(B, Z * C, Y, X)
    return out


def voxel_pooling_grad(geom, grad_out, dx, bx, nx, C):
    """Backward of voxel_pooling wrt the volume x (vt_mine.py:49-56 composed
    with the gather/scatter of :161-176): every kept point receives the
    gradient of its voxel, dropped points receive 0.
    grad_out f32[B, C*Z, Y, X] -> f32[B,N,D,H,W,C]."""
    B, N, D, H, W, _ = geom.shape
    Np = B * N * D * H * W
    idx, kept = voxel_index(geom.reshape(Np, 3), dx, bx, nx)
    nxl = nx.astype(np.int64)
    X, Y, Z = int(nxl[0]), int(nxl[1]), int(nxl[2])
    g = grad_out.reshape(B, Z, C, Y, X)
    batch_ix = np.repeat(np.arange(B, dtype=np.int64), Np // B)
    gx = np.zeros((Np, C), dtype=f32)
    ii = idx[kept]
    bb = batch_ix[kept]
    gx[kept] = g[bb, ii[:, 2], :, ii[:, 1], ii[:, 0]]
    return gx.reshape(B, N, D, H, W, C)


# --------------------------------------------------------------------------
# bev_pool extension contract: ops/bev_pool/bev_pool.py + src/bev_pool_cuda.cu
# --------------------------------------------------------------------------
def bev_pool_prepare(feats, coords, B, D, H, W):
    """bev_pool.py:83-97 + :39-46.  coords int[n,4] = (x, y, z, b).
    Returns (feats_sorted, coords_sorted int32, interval_starts int32,
    interval_lengths int32).  argsort is *stable* here (the reference's is
    unspecified); any order within a rank run is a valid reference order."""
    coords = np.asarray(coords, dtype=np.int64)
    ranks = (coords[:, 0] * (W * D * B) + coords[:, 1] * (D * B)
             + coords[:, 2] * B + coords[:, 3])
    order = np.argsort(ranks, kind="stable")
    feats, coords, ranks = feats[order], coords[order], ranks[order]
    n = feats.shape[0]
    kept = np.ones(n, dtype=bool)
    kept[1:] = ranks[1:] != ranks[:-1]
    starts = np.flatnonzero(kept).astype(np.int32)
    lengths = np.zeros_like(starts)
    if starts.shape[0] > 0:
        lengths[:-1] = starts[1:] - starts[:-1]
        lengths[-1] = n - starts[-1]
    return feats, coords.astype(np.int32), starts, lengths


def bev_pool_forward(x, geom, starts, lengths, B, D, H, W, exact=False):
    """src/bev_pool_cuda.cu:20-42.  out[b, z, x, y, c] = sum_i x[start+i, c],
    fp32 sequential (exact=False, the reference kernel's own order) or fp64
    (exact=True).  Output f32[B, D, H, W, C], zero where no interval lands."""
    n, C = x.shape
    out = np.zeros((B, D, H, W, C), dtype=f32)
    for s, l in zip(starts, lengths):
        gx, gy, gz, gb = geom[s]
        if exact:
            out[gb, gz, gx, gy] = x[s:s + l].astype(np.float64).sum(0).astype(f32)
        else:
            acc = np.zeros(C, dtype=f32)
            for r in range(s, s + l):
                acc = (acc + x[r]).astype(f32)
            out[gb, gz, gx, gy] = acc
    return out


def bev_pool_backward(out_grad, geom, starts, lengths, n):
    """src/bev_pool_cuda.cu:61-84.  x_grad[start+i, :] = out_grad[b, z, x, y, :]."""
    C = out_grad.shape[-1]
    xg = np.zeros((n, C), dtype=f32)
    for s, l in zip(starts, lengths):
        gx, gy, gz, gb = geom[s]
        xg[s:s + l] = out_grad[gb, gz, gx, gy]
    return xg


def bev_pool(feats, coords, B, D, H, W, exact=False):
    """bev_pool.py:83-97 end to end -> f32[B, C, D, H, W]."""
    f, g, st, ln = bev_pool_prepare(feats, coords, B, D, H, W)
    out = bev_pool_forward(f, g, st, ln, B, D, H, W, exact=exact)
    return np.ascontiguousarray(out.transpose(0, 4, 1, 2, 3))


# --------------------------------------------------------------------------
# lift: vt_mine.py:333-335 / bevdet_distill_more.py:411-416
# --------------------------------------------------------------------------
def lift(depth_prob, img_feat, B, N):
    """volume[b,n,d,h,w,c] = depth_prob[bn,d,h,w] * img_feat[bn,c,h,w]."""
    BN, D, H, W = depth_prob.shape
    C = img_feat.shape[1]
    vol = (depth_prob[:, None] * img_feat[:, :, None]).astype(f32)  # [BN,C,D,H,W]
    vol = vol.reshape(B, N, C, D, H, W).transpose(0, 1, 3, 4, 5, 2)
    return np.ascontiguousarray(vol)


def lift_splat(depth_prob, img_feat, geom, dx, bx, nx, exact=True):
    """lift followed by voxel_pooling (the fused op's reference sequence)."""
    B, N = geom.shape[:2]
    return voxel_pooling(geom, lift(depth_prob, img_feat, B, N), dx, bx, nx, exact=exact)


def lift_splat_grad(depth_prob, img_feat, geom, grad_out, dx, bx, nx):
    """Gradients of lift_splat wrt depth_prob [BN,D,H,W] and img_feat [BN,C,H,W]
    (fp64 accumulate): g_vol = grad of voxel; g_depth = sum_c g_vol*feat,
    g_feat = sum_d g_vol*depth."""
    B, N = geom.shape[:2]
    BN, D, H, W = depth_prob.shape
    C = img_feat.shape[1]
    gv = voxel_pooling_grad(geom, grad_out, dx, bx, nx, C).astype(np.float64)  # B,N,D,H,W,C
    gv = gv.reshape(BN, D, H, W, C)
    feat = img_feat.astype(np.float64).transpose(0, 2, 3, 1)  # BN,H,W,C
    g_depth = np.einsum("ndhwc,nhwc->ndhw", gv, feat)
    g_feat = np.einsum("ndhwc,ndhw->nchw", gv, depth_prob.astype(np.float64))
    return g_depth.astype(f32), g_feat.astype(f32)
