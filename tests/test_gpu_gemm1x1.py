"""1x1-convolution GEMM kernels (csrc/gemm1x1.hip) vs torch in fp64: forward (+ statistics rows), data gradient, weight gradient; both
column-tile widths, strided channel counts, every accumulator-tile combination of the weight gradient.  fp32 FMA chains: 2e-6 of the
output scale (the library's direct kernel measured beside it)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

SHAPES = [(2, 64, 64, 8, 16), (2, 64, 128, 16, 16), (1, 256, 64, 16, 8), (3, 128, 512, 16, 8), (1, 1024, 256, 8, 16), (2, 96, 192, 8, 8)]


def _mk(N, C, Co, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, C, H, W), generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Co, C, 1, 1), generator=g) / C ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    return x, w


@pytest.mark.parametrize("N,C,Co,H,W", SHAPES)
def test_forward_statistics_and_gradients_vs_fp64(N, C, Co, H, W):
    from distill_bev_amd import gemm1x1 as G
    x, w = _mk(N, C, Co, H, W, 3)
    assert G.eligible(x, w)
    y, part = G.conv1x1_stats(x, w)
    ref = F.conv2d(x.double(), w.double())
    scale = float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= 2e-6 * scale
    assert y.is_contiguous(memory_format=torch.channels_last)
    s = part.double().sum(0)
    assert torch.allclose(s[0], y.double().sum((0, 2, 3)), rtol=1e-5, atol=1e-4 * scale)
    assert torch.allclose(s[1], (y.double() ** 2).sum((0, 2, 3)), rtol=1e-5, atol=1e-4 * scale)
    y2, part2 = G.conv1x1_stats(x, w)
    assert torch.equal(y, y2) and torch.equal(part, part2)
    assert torch.equal(G.conv1x1(x, w), y)
    # gradients
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(4)).to(DEV).contiguous(memory_format=torch.channels_last)
    gx, gw = torch.autograd.grad(G.conv1x1(xr, wr), (xr, wr), gy)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    rx, rw = torch.autograd.grad(F.conv2d(xd, wd), (xd, wd), gy.double())
    assert float((gx.double() - rx).abs().max()) <= 3e-6 * float(rx.abs().max())
    assert float((gw.double() - rw).abs().max()) <= 3e-6 * float(rw.abs().max())
    assert gw.shape == w.shape
    gx2, gw2 = torch.autograd.grad(G.conv1x1(xr, wr), (xr, wr), gy)
    assert torch.equal(gx, gx2) and torch.equal(gw, gw2)


def test_ineligible_shapes_keep_the_stock_convolution(monkeypatch):
    from distill_bev_amd import gemm1x1 as G
    monkeypatch.setattr(G, "_ON", True)
    ok = G.GemmConv2d(64, 64, 1, bias=False).to(DEV)
    xo = torch.randn(2, 64, 8, 8, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yo = ok(xo)
    assert type(yo.grad_fn).__name__ == "_Conv1x1Backward" and torch.allclose(yo, F.conv2d(xo, ok.weight), atol=1e-5)
    conv = G.GemmConv2d(64, 59, 1, bias=False).to(DEV)
    x = torch.randn(2, 64, 8, 8, device=DEV).contiguous(memory_format=torch.channels_last)
    assert not G.eligible(x, conv.weight)
    assert torch.allclose(conv(x), F.conv2d(x, conv.weight), atol=1e-6)
    x2 = torch.randn(1, 64, 5, 7, device=DEV).contiguous(memory_format=torch.channels_last)         # 35 pixels
    assert not G.eligible(x2, torch.empty(64, 64, 1, 1, device=DEV))
