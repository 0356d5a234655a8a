"""The 3x3 / stride-2 / padding-1 convolution (conv2 of a stage-first ResNet bottleneck) as an implicit bf16x6 GEMM (csrc/gemm_bf6.hip,
b6_fwd2<.., CONV>): forward + statistics against the fp64 convolution and the library's fp32 one, the module class with the library's
gradients, fall-backs, and the one-launch re-pack of its filter."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(2, 64, 16, 32, 128), (4, 128, 16, 16, 128), (1, 256, 32, 16, 64), (6, 128, 64, 176, 128),
          (3, 128, 58, 100, 128), (1, 64, 10, 6, 64), (5, 256, 14, 18, 192)]      # round 6: N H/2 W/2 no multiple of 128


def _data(N, C, H, W, Co, seed=0):
    g = torch.Generator().manual_seed(seed)
    dev = torch.device("cuda:0")
    x = torch.randn((N, C, H, W), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Co, C, 3, 3), generator=g) / (3.0 * C ** 0.5)).to(dev).contiguous(memory_format=torch.channels_last)
    return x, w


@pytest.mark.parametrize("N,C,H,W,Co", SHAPES)
def test_forward_and_statistics(N, C, H, W, Co, monkeypatch):
    from distill_bev_amd import gemm_bf6 as G
    monkeypatch.setattr(G, "_MIN_ITEMS", 1)
    x, w = _data(N, C, H, W, Co)
    assert G.eligible_c3s2(x, w)
    y, part = G.product_c3s2(x, w, stats=True)
    assert torch.equal(y, G.product_c3s2(x, w)) and y.is_contiguous(memory_format=torch.channels_last)
    ref64 = F.conv2d(x.double(), w.double(), None, 2, 1)
    lib = F.conv2d(x, w, None, 2, 1)
    assert y.shape == ref64.shape
    scale = float(ref64.abs().max())
    err, err_lib = float((y.double() - ref64).abs().max()) / scale, float((lib.double() - ref64).abs().max()) / scale
    assert err <= max(1.5 * err_lib, 6e-7), (err, err_lib)       # the bf16x6 contract: no worse than the library's fp32 kernels
    s = part.double().sum(0)
    yy = y.double().permute(0, 2, 3, 1).reshape(-1, Co)
    assert torch.allclose(s[0], yy.sum(0), rtol=1e-5, atol=1e-3) and torch.allclose(s[1], (yy * yy).sum(0), rtol=1e-5, atol=1e-3)


def test_edges_are_zero_padding_not_neighbouring_rows():
    """an input that is non-zero only on the border rows / columns: every output value comes through a tap at the map's edge"""
    from distill_bev_amd import gemm_bf6 as G
    G_min = G._MIN_ITEMS
    G._MIN_ITEMS = 1
    try:
        x, w = _data(2, 64, 16, 32, 64, seed=3)
        m = torch.zeros_like(x)
        m[:, :, 0, :] = 1; m[:, :, -1, :] = 1; m[:, :, :, 0] = 1; m[:, :, :, -1] = 1
        x = (x * m).contiguous(memory_format=torch.channels_last)
        y = G.product_c3s2(x, w)
        ref = F.conv2d(x.double(), w.double(), None, 2, 1)
        assert float((y.double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    finally:
        G._MIN_ITEMS = G_min


def test_module_class_gradients_and_fallbacks(monkeypatch):
    from distill_bev_amd import _lib as L
    from distill_bev_amd import gemm_bf6 as G
    monkeypatch.setattr(G, "_MIN_ITEMS", 1)
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    conv = nn.Conv2d(128, 128, 3, 2, 1, bias=False).to(dev)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    other = nn.Sequential(nn.Conv2d(128, 128, 3, 2, 1, bias=True), nn.Conv2d(96, 128, 3, 2, 1, bias=False), nn.Conv2d(128, 128, 3, 1, 1, bias=False))
    assert G.use_bf6_convs(nn.Sequential(conv)) == 1 and type(conv) is G.Bf6Conv3x3S2 and G.use_bf6_convs(other) == 0
    x = torch.randn((4, 128, 16, 32), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    L.kernel_timing(["b6_fwd"])
    y = conv(x)
    torch.cuda.synchronize()
    assert len(L.kernel_timing_read().get("b6_fwd", [])) == 1
    L.kernel_timing(False)
    gy = torch.randn_like(y)
    gx, gw = torch.autograd.grad(y, [x, conv.weight], gy)
    ref = F.conv2d(x, conv.weight, None, 2, 1)
    rgx, rgw = torch.autograd.grad(ref, [x, conv.weight], gy)
    assert torch.allclose(y, ref, rtol=1e-4, atol=1e-5)
    assert torch.equal(gx, rgx) and torch.equal(gw, rgw)          # the library's own gradients
    with torch.no_grad():
        assert torch.equal(conv(x), y)
    # fall-backs: NCHW input, odd map, a filter that is not channels-last
    xn = x.detach().contiguous()
    assert not G.eligible_c3s2(xn, conv.weight) and torch.allclose(conv(xn), F.conv2d(xn, conv.weight, None, 2, 1), atol=1e-5)
    xo = torch.randn((4, 128, 15, 32), device=dev).contiguous(memory_format=torch.channels_last)
    assert not G.eligible_c3s2(xo, conv.weight) and conv(xo).shape == (4, 128, 8, 16)
    wn = conv.weight.detach().contiguous()
    assert not G.eligible_c3s2(x, wn)


def test_one_launch_repack_of_the_filter_matches_the_lazy_pack(monkeypatch):
    from distill_bev_amd import gemm_bf6 as G
    from distill_bev_amd.packer import WeightPacker
    monkeypatch.setattr(G, "_MIN_ITEMS", 1)
    dev = torch.device("cuda:0")
    x = torch.randn((4, 64, 16, 32), device=dev).contiguous(memory_format=torch.channels_last)
    outs = {}
    for use_packer in (False, True):
        torch.manual_seed(7)
        net = nn.Sequential(nn.Conv2d(64, 128, 3, 2, 1, bias=False), nn.ReLU(), nn.Conv2d(128, 64, 1, bias=False)).to(dev)
        for m in net.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        assert G.use_bf6_convs(net) == 2
        opt = torch.optim.AdamW(net.parameters(), lr=1e-2, fused=True)
        packer = WeightPacker([net]) if use_packer else None
        ys = []
        for _ in range(3):
            y = net(x)
            opt.zero_grad(set_to_none=True)
            y.square().mean().backward()
            opt.step()
            if packer is not None:
                packer.repack()
                w = net[0].weight
                assert w._dbev_bf6_packs[0][0] == w._version     # found fresh by the next forward
            ys.append(y.detach().clone())
        if packer is not None:
            assert len(packer.bf6) == 2 and packer.launches == 3
        outs[use_packer] = ys
    for a, b in zip(outs[False], outs[True]):
        assert torch.equal(a, b)
