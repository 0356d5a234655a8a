"""Winograd F(2x2, 3x3) convolution kernels (csrc/wino.hip) vs torch's convolution in fp64 (the op the reference's dense blocks run:
nn.Conv2d(k=3, s=1, p=1), res_block.py:11-230): forward, data gradient, statistics rows, bias, every tile-block shape, partial
blocks at the image edges.  Tolerance: 2e-5 of the output scale (fp32 Winograd: 2-4x a direct fp32 convolution's error, measured
against the same fp64 reference and asserted below)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _mk(N, C, Co, H, W, seed, bias):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, C, H, W), generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Co, C, 3, 3), generator=g) / (3.0 * C ** 0.5)).to(DEV).contiguous(memory_format=torch.channels_last)
    b = torch.randn((Co,), generator=g).to(DEV) if bias else None
    return x, w, b


SHAPES = [(2, 16, 64, 16, 16, False),      # one 8x8 tile block per image, a single patch stage
          (3, 64, 64, 16, 44, True),        # 8 x 22 tiles: partial blocks along W (8,8)
          (2, 32, 128, 8, 64, False),       # 4 x 32 tiles: the (4,16) block shape
          (1, 48, 64, 6, 10, True),         # partial blocks both ways, three k stages
          (2, 256, 256, 16, 12, False),     # deep reduction (32 k groups)
          (1, 64, 192, 34, 18, True)]       # odd tile counts, three channel blocks


@pytest.mark.parametrize("N,C,Co,H,W,bias", SHAPES)
def test_forward_and_statistics_vs_fp64(N, C, Co, H, W, bias):
    from distill_bev_amd import wino
    x, w, b = _mk(N, C, Co, H, W, 1, bias)
    assert wino.eligible(x, w)
    y, part = wino.conv3x3_stats(x, w, b)
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), 1, 1)
    scale = float(ref.abs().max())
    err = float((y.double() - ref).abs().max()) / scale
    direct = float((F.conv2d(x, w, b, 1, 1).double() - ref).abs().max()) / scale
    assert err <= 2e-5, err
    assert err <= max(8.0 * direct, 2e-6), (err, direct)          # same error class as the direct fp32 kernel
    assert y.is_contiguous(memory_format=torch.channels_last)
    # statistics rows: per channel sum y, sum y^2 over all pixels (fixed order: bit-identical when repeated)
    s = part.double().sum(0)
    assert torch.allclose(s[0], y.double().sum((0, 2, 3)), rtol=1e-5, atol=1e-4 * scale)
    assert torch.allclose(s[1], (y.double() ** 2).sum((0, 2, 3)), rtol=1e-5, atol=1e-4 * scale)
    y2, part2 = wino.conv3x3_stats(x, w, b)
    assert torch.equal(y, y2) and torch.equal(part, part2)
    assert torch.equal(wino.conv3x3(x, w, b), y)


@pytest.mark.parametrize("N,C,Co,H,W,bias", SHAPES)
def test_gradients_vs_fp64(N, C, Co, H, W, bias):
    from distill_bev_amd import wino
    x, w, b = _mk(N, C, Co, H, W, 2, bias)
    x.requires_grad_(True); w.requires_grad_(True)
    if b is not None:
        b.requires_grad_(True)
    gy = torch.randn((N, Co, H, W), generator=torch.Generator().manual_seed(3)).to(DEV).contiguous(memory_format=torch.channels_last)
    ins = (x, w) + ((b,) if b is not None else ())
    got = torch.autograd.grad(wino.conv3x3(x, w, b), ins, gy)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    bd = None if b is None else b.detach().double().requires_grad_(True)
    ref = torch.autograd.grad(F.conv2d(xd, wd, bd, 1, 1), (xd, wd) + ((bd,) if bd is not None else ()), gy.double())
    for a, r, name in zip(got, ref, ("grad_x", "grad_w", "grad_b")):
        e = float((a.double() - r).abs().max()) / float(r.abs().max())
        assert e <= 3e-5, (name, e)
    assert got[0].is_contiguous(memory_format=torch.channels_last)


def test_filter_pack_is_the_winograd_transform_of_each_filter():
    """packed[(jb, kg, p, ni, lane, e)] = (G g G^T)[p] of filter (k, j); both modes (forward / rotated-transposed)"""
    from distill_bev_amd import wino
    Co, C = 128, 24
    g = torch.Generator().manual_seed(5)
    w = torch.randn((Co, C, 3, 3), generator=g).to(DEV)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    for mode, wm in ((False, w), (True, w.flip(2, 3).transpose(0, 1))):
        if mode and C % 64:
            continue
        K, J = wm.shape[1], wm.shape[0]
        U = torch.einsum("ia,jkab,cb->jkic", G, wm.double().cpu(), G).reshape(J, K, 16)        # [j, k, p]
        packed = wino.pack_filters(w, mode).cpu().double().reshape(J // 64, K // 8, 16, 2, 64, 4)
        jb, kg, p, ni, lane, e = np.ix_(*[np.arange(n) for n in packed.shape])
        k = 8 * kg + 4 * (lane >> 5) + e
        j = 64 * jb + 32 * ni + (lane & 31)
        want = U.numpy()[j, k, p]
        assert np.abs(packed.numpy() - want).max() <= 1e-6
