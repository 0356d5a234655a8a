"""The shipped MIOpen solver tables (distill_bev_amd/miopen_db) pick fp32 kernels: the heaviest convolution problems of the step,
run with the tables active, agree with fp64 arithmetic to fp32 rounding: forward, data gradient and weight gradient element by
element at sampled positions against fp64 sums of the defining products.  (A kernel computing in a reduced-precision matrix
format would miss these bounds by 2-3 orders of magnitude.)"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_tables_are_shipped_and_private_per_process():
    from distill_bev_amd import miopen_tuning as MT
    files = sorted(os.listdir(MT._DB))
    assert any(f.endswith(".ufdb.txt") for f in files) and any(f.endswith(".udb.txt") for f in files)
    if os.environ.get("DBEV_MIOPEN_DB", "1") == "0":
        assert MT.use_shipped_db() is None
        return
    path = MT.use_shipped_db()
    if path is None:                       # a user-set MIOPEN_USER_DB_PATH wins
        assert os.environ.get("MIOPEN_USER_DB_PATH")
        return
    assert os.environ["MIOPEN_USER_DB_PATH"] == path and path != MT._DB
    assert sorted(os.listdir(path))[:2] == files[:2]
    assert MT.use_shipped_db() == path     # idempotent


@pytest.mark.parametrize("N,C,K,H,W,k,s", [
    (48, 256, 256, 16, 44, 3, 1),     # stage 3 + depth net: the largest share of the step's convolution time
    (8, 512, 256, 128, 128, 3, 1),    # BEV neck
    (8, 640, 512, 64, 64, 3, 1),      # BEV neck
    (48, 64, 256, 64, 176, 1, 1),     # stage 1 expand (HBM-bound)
    (48, 3, 64, 256, 704, 7, 2),      # stem
])
def test_tuned_kernels_compute_in_fp32(N, C, K, H, W, k, s):
    from distill_bev_amd.miopen_tuning import use_shipped_db
    use_shipped_db()
    g = torch.Generator(device=DEV).manual_seed(N + C + K)
    pad = k // 2
    x = torch.randn((N, C, H, W), device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((K, C, k, k), device=DEV, generator=g) / (C * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
    y = F.conv2d(x, w, None, s, pad)
    Ho, Wo = y.shape[2:]
    # forward: 64 sampled output pixels, all K channels, fp64 dot products over the zero-padded patches
    xp = F.pad(x, (pad, pad, pad, pad)).double()
    wd = w.double()
    idx = torch.randint(0, N * Ho * Wo, (64,), generator=torch.Generator().manual_seed(1)).tolist()
    worst = 0.0
    for i in idx:
        n, r = divmod(i, Ho * Wo)
        h, ww = divmod(r, Wo)
        patch = xp[n, :, h * s:h * s + k, ww * s:ww * s + k]
        ref = (wd * patch.unsqueeze(0)).sum(dim=(1, 2, 3))
        worst = max(worst, float((y[n, :, h, ww].double() - ref).abs().max() / ref.abs().max()))
    assert worst <= 2e-5, worst
    # gradients, element by element in fp64 at sampled positions
    dy = torch.randn(y.shape, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    need_dx = C > 3
    gx, gw, _ = torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [need_dx, True, False])
    dyd = dy.double()
    rs = torch.Generator().manual_seed(2)
    #   dw[k, c, r, q] = sum_{n, ho, wo} dy[n, k, ho, wo] * xpad[n, c, ho * s + r, wo * s + q]
    worst_w, scale_w = 0.0, float(gw.abs().max())
    for _ in range(16):
        kk, c, r, q = (int(torch.randint(0, m, (1,), generator=rs)) for m in (K, C, k, k))
        ref = float((dyd[:, kk] * xp[:, c, r:r + (Ho - 1) * s + 1:s, q:q + (Wo - 1) * s + 1:s]).sum())
        worst_w = max(worst_w, abs(float(gw[kk, c, r, q]) - ref) / scale_w)
    assert worst_w <= 2e-5, worst_w               # fp32 accumulation over N*Ho*Wo = 3e4 .. 1e6 products
    #   dx[n, c, h, w] = sum_{k, r, q} dy[n, k, (h + pad - r) / s, (w + pad - q) / s] * w[k, c, r, q]   (integer, in range)
    if need_dx:
        worst_x, scale_x = 0.0, float(gx.abs().max())
        for _ in range(32):
            n, c, h, ww = (int(torch.randint(0, m, (1,), generator=rs)) for m in (N, C, H, W))
            ref = 0.0
            for r in range(k):
                for q in range(k):
                    a, b = h + pad - r, ww + pad - q
                    if a % s or b % s or not (0 <= a // s < Ho and 0 <= b // s < Wo):
                        continue
                    ref += float((dyd[n, :, a // s, b // s] * wd[:, c, r, q]).sum())
            worst_x = max(worst_x, abs(float(gx[n, c, h, ww]) - ref) / scale_x)
        assert worst_x <= 2e-5, worst_x
