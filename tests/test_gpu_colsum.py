"""Bias terms of the library convolutions (csrc/colsum.hip, distill_bev_amd/colsum.py) vs the torch ops the reference runs
(nn.Conv2d(bias=True) -> ATen: bias added behind the convolution, grad_bias = grad_output.sum((0, 2, 3)); necks/fpn.py:77-95,
view_transformer_mine.py:288-309, bevdet_distill.py:99-132).  channel_sum against an fp64 sum at 2e-6 of the column's absolute sum
(fp32 accumulation in a fixed order), bit-reproducible; the module swaps against the stock modules."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("N,C,H,W", [(48, 59, 16, 44), (48, 27, 16, 44), (8, 256, 128, 128), (3, 64, 5, 7), (1, 1, 1, 1), (2, 3, 1, 5),
                                     (1, 2304, 6, 4), (7, 260, 9, 11), (1, 512, 1, 1), (48, 512, 8, 22)])
def test_channel_sum_vs_fp64(N, C, H, W):
    from distill_bev_amd.colsum import channel_sum
    g = torch.Generator().manual_seed(N * 1000 + C)
    t = torch.randn((N, C, H, W), generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    out = channel_sum(t)
    ref = t.double().sum((0, 2, 3))
    scale = t.double().abs().sum((0, 2, 3)).clamp(min=1e-30)
    assert out.shape == (C,) and out.dtype == torch.float32
    assert float(((out.double() - ref).abs() / scale).max()) < 2e-6
    assert torch.equal(out, channel_sum(t))                  # fixed summation order
    # the [M, C] view of the same memory
    assert torch.equal(out, channel_sum(t.permute(0, 2, 3, 1).reshape(-1, C)))


def test_channel_sum_rejects_other_layouts():
    from distill_bev_amd import _lib as L
    from distill_bev_amd.colsum import channel_sum
    with pytest.raises(L.DbevHipError):
        channel_sum(torch.randn((2, 8, 4, 4), device=DEV))               # NCHW memory
    with pytest.raises(L.DbevHipError):
        channel_sum(torch.randn((2, 8, 4, 4)))                           # host tensor: no CPU fallback in the product path


@pytest.mark.parametrize("ci,co,k,s,p", [(64, 96, 1, 1, 0), (32, 27, 3, 1, 1), (48, 64, 3, 2, 1), (16, 59, 1, 1, 0)])
def test_bias_sum_conv_matches_the_stock_module(ci, co, k, s, p):
    from distill_bev_amd.colsum import BiasSumConv2d, use_bias_sum_convs
    torch.manual_seed(3)
    ref = nn.Conv2d(ci, co, k, s, p).to(DEV).to(memory_format=torch.channels_last)
    mod = nn.Sequential(nn.Conv2d(ci, co, k, s, p)).to(DEV).to(memory_format=torch.channels_last)
    mod[0].load_state_dict(ref.state_dict())
    assert use_bias_sum_convs(mod) == 1 and type(mod[0]) is BiasSumConv2d and use_bias_sum_convs(mod) == 0
    assert list(mod[0].state_dict()) == list(ref.state_dict())
    x = torch.randn((4, ci, 12, 20), device=DEV).contiguous(memory_format=torch.channels_last)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = ref(xa), mod(xb)
    assert torch.equal(ya, yb)                                # same kernel, same separate bias pass
    gy = torch.randn_like(ya)
    ya.backward(gy); yb.backward(gy)
    assert torch.equal(xa.grad, xb.grad) and torch.equal(ref.weight.grad, mod[0].weight.grad)
    gb64 = gy.double().sum((0, 2, 3))
    assert float((mod[0].bias.grad.double() - gb64).abs().max()) < 2e-6 * float(gy.double().abs().sum((0, 2, 3)).max())
    with torch.no_grad():                                     # no graph: the stock path
        assert torch.equal(mod(x), ref(x))


def test_bias_in_front_of_a_training_norm_cancels():
    """ThreeLayer / TwoLayer of the adaptation layers (bevdet_distill.py:70-132): conv(bias) -> BatchNorm2d (training) -> ReLU"""
    from distill_bev_amd.detectors import ThreeLayer
    torch.manual_seed(5)
    ref = nn.Sequential(nn.Conv2d(64, 64, 1), nn.BatchNorm2d(64), nn.ReLU(), nn.Conv2d(64, 64, 1), nn.BatchNorm2d(64), nn.ReLU(),
                        nn.Conv2d(64, 128, 1), nn.BatchNorm2d(128), nn.ReLU()).to(DEV).double()
    m = ThreeLayer(64, 64, 128, kernel_size=1, stride=1).to(DEV).to(memory_format=torch.channels_last)
    with torch.no_grad():
        for i, (c, n) in enumerate(((m.conv1, m.norm1), (m.conv2, m.norm2), (m.conv3, m.norm3))):
            c.bias.normal_(0, 0.5); n.weight.uniform_(0.5, 1.5); n.bias.normal_(0, 0.2)
            ref[3 * i].weight.copy_(c.weight.double()); ref[3 * i].bias.copy_(c.bias.double())
            ref[3 * i + 1].weight.copy_(n.weight.double()); ref[3 * i + 1].bias.copy_(n.bias.double())
    x = torch.randn((4, 64, 16, 24), device=DEV).contiguous(memory_format=torch.channels_last)
    xa = x.double().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ya, yb = ref(xa), m(xb)
    assert float((ya - yb.double()).abs().max()) < 2e-5
    gy = torch.randn_like(yb)
    ya.backward(gy.double()); yb.backward(gy)
    assert float((xa.grad - xb.grad.double()).abs().max()) < 2e-5 * max(1.0, float(xa.grad.abs().max()))
    for i, (c, n) in enumerate(((m.conv1, m.norm1), (m.conv2, m.norm2), (m.conv3, m.norm3))):
        rc, rn = ref[3 * i], ref[3 * i + 1]
        assert float((rc.weight.grad - c.weight.grad.double()).abs().max()) < 2e-5 * max(1.0, float(rc.weight.grad.abs().max()))
        assert float((rn.weight.grad - n.weight.grad.double()).abs().max()) < 5e-5 * max(1.0, float(rn.weight.grad.abs().max()))
        assert float((rn.bias.grad - n.bias.grad.double()).abs().max()) < 5e-5 * max(1.0, float(rn.bias.grad.abs().max()))
        # the bias gradient is a sum of the norm's input gradient: zero up to rounding in the reference, exactly zero here
        assert float(rc.bias.grad.abs().max()) < 1e-9 and float(c.bias.grad.abs().max()) == 0.0
        # running statistics track mean(conv) + bias
        assert float((rn.running_mean - n.running_mean.double()).abs().max()) < 1e-5
        assert float((rn.running_var - n.running_var.double()).abs().max()) < 1e-5
        assert int(n.num_batches_tracked) == 1
    m.eval(); ref.eval()                                      # evaluation mode: the bias is applied as usual
    with torch.no_grad():
        assert float((ref(x.double()) - m(x).double()).abs().max()) < 2e-5
