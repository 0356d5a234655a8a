"""csrc/stem.hip: the ResNet stem convolution (3 -> 64, 7x7, stride 2, padding 3, no bias; mmdet ResNet `conv1`) on the fp32 matrix
cores -- forward (+ the norm's statistics in the epilogue) and weight gradient against the fp64 convolution and the library's fp32
one, on full tiles, ragged edges and tiny maps; the module path (`StemConv2d`, `conv_norm_relu_max_pool`) against the module sequence."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(2, 64, 128), (3, 37, 45), (1, 7, 7), (2, 256, 704), (5, 9, 200)]


def _data(N, H, W, seed=0):
    g = torch.Generator().manual_seed(seed)
    dev = torch.device("cuda:0")
    x = torch.randn((N, 3, H, W), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((64, 3, 7, 7), generator=g) * 0.1).to(dev).contiguous(memory_format=torch.channels_last)
    return x, w


@pytest.mark.parametrize("N,H,W", SHAPES)
def test_forward_and_statistics_against_fp64_and_the_library(N, H, W):
    from distill_bev_amd import stem
    x, w = _data(N, H, W)
    z, part = stem.stem_conv_stats(x, w)
    z2 = stem.stem_conv(x, w)
    assert torch.equal(z, z2) and z.is_contiguous(memory_format=torch.channels_last)
    ref64 = F.conv2d(x.double(), w.double(), None, 2, 3)
    lib = F.conv2d(x, w, None, 2, 3)
    assert z.shape == ref64.shape
    err = float((z.double() - ref64).abs().max())
    err_lib = float((lib.double() - ref64).abs().max())
    assert err <= max(2.0 * err_lib, 2e-6 * float(ref64.abs().max())), (err, err_lib)
    # the statistics rows: channel sums of z and z^2 over all pixels
    s = part.double().sum(0)
    zz = z.double().permute(0, 2, 3, 1).reshape(-1, 64)
    assert torch.allclose(s[0], zz.sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[1], (zz * zz).sum(0), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("N,H,W", SHAPES)
def test_weight_gradient_against_fp64_and_the_library_and_is_bit_reproducible(N, H, W):
    from distill_bev_amd import stem
    x, w = _data(N, H, W, seed=1)
    w.requires_grad_(True)
    z = stem.stem_conv(x, w)
    gz = torch.randn(z.shape, generator=torch.Generator().manual_seed(2)).to(z.device).contiguous(memory_format=torch.channels_last)
    (gw,) = torch.autograd.grad(z, w, gz, retain_graph=True)
    (gw2,) = torch.autograd.grad(z, w, gz)
    assert torch.equal(gw, gw2) and gw.stride() == w.stride()
    ref64 = torch.ops.aten.convolution_backward(gz.double(), x.double(), w.detach().double(), None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1,
                                                [False, True, False])[1]
    lib = torch.ops.aten.convolution_backward(gz, x, w.detach(), None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    err = float((gw.double() - ref64).abs().max())
    err_lib = float((lib.double() - ref64).abs().max())
    assert err <= max(2.0 * err_lib, 2e-6 * float(ref64.abs().max())), (err, err_lib)


def test_module_path_matches_the_module_sequence_and_falls_back():
    from distill_bev_amd import stem
    from distill_bev_amd.pool import conv_norm_relu_max_pool
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False).to(dev)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    holder = nn.Sequential(conv)
    assert stem.use_stem_convs(holder) == 1 and stem.use_stem_convs(holder) == 1 and type(conv) is stem.StemConv2d
    norm, pool = nn.BatchNorm2d(64).to(dev).train(), nn.MaxPool2d(3, 2, 1)
    norm_ref = nn.BatchNorm2d(64).to(dev).train()
    x = torch.randn((4, 3, 64, 96), device=dev).contiguous(memory_format=torch.channels_last)
    from distill_bev_amd import _lib as L
    L.kernel_timing(True)
    y = conv_norm_relu_max_pool(conv, norm, pool, x)
    torch.cuda.synchronize()
    ran = set(L.kernel_timing_read())
    L.kernel_timing(False)
    assert "stem_fwd" in ran and "bn_stats" not in ran, ran
    ref = pool(torch.relu(norm_ref(F.conv2d(x, conv.weight, None, 2, 3))))
    assert torch.allclose(y, ref, rtol=1e-4, atol=1e-4)
    assert torch.allclose(norm.running_mean, norm_ref.running_mean, rtol=1e-4, atol=1e-5)
    assert torch.allclose(norm.running_var, norm_ref.running_var, rtol=1e-4, atol=1e-5)
    gy = torch.randn_like(y)
    g = torch.autograd.grad(y, [conv.weight, norm.weight, norm.bias], gy)
    gr = torch.autograd.grad(ref, [conv.weight, norm_ref.weight, norm_ref.bias], gy)
    for a, b in zip(g, gr):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max())), float((a - b).abs().max())
    # not the stem's geometry / layout / an input with a gradient: the module's own convolution
    xn = x.contiguous()
    assert not stem.eligible(conv, xn) and torch.allclose(conv(xn), F.conv2d(xn, conv.weight, None, 2, 3))
    xg = x.clone().requires_grad_(True)
    assert not stem.eligible(conv, xg)
    out = conv(xg)
    (gx,) = torch.autograd.grad(out.sum(), xg)
    assert gx.shape == xg.shape
    other = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=True), nn.Conv2d(3, 32, 7, 2, 3, bias=False), nn.Conv2d(3, 64, 7, 1, 3, bias=False))
    assert stem.use_stem_convs(other) == 0


def test_rejects_geometry_it_does_not_take():
    from distill_bev_amd import _lib as L
    assert L.lib().dbev_stem7x7s2_workspace_bytes(1, 6, 64) == 0 and L.lib().dbev_stem7x7s2_stats_rows(0, 64, 64) == 0
    assert L.lib().dbev_stem7x7s2_workspace_bytes(48, 256, 704) > 0
