"""3x3 / s1 / p1 convolutions with 1-3 output channels (csrc/skinny_conv.hip) through the C ABI vs F.conv2d in fp64 on the host:
output, grad_x, grad_weight, grad_bias.  fp32 kernels, tolerances relative to the tensor scale."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("N,Ci,H,W,Co,bias", [
    (2, 64, 128, 128, 3, True),      # the 'dim' branch of the head
    (8, 64, 128, 128, 1, True),      # 'height' / single-class heat map at the step's batch size
    (3, 64, 7, 5, 2, False),         # tiny map: every pixel is a border pixel for some tap
    (1, 32, 9, 16, 3, True),         # Cin/4 = 8 lanes per pixel
    (2, 256, 6, 10, 2, True),        # Cin/4 = 64 lanes per pixel
    (1, 128, 1, 1, 1, True),         # 1x1 image: only the centre tap is inside
])
def test_forward_backward_vs_fp64_conv2d(N, Ci, H, W, Co, bias):
    from distill_bev_amd.skinny_conv import eligible, skinny_conv3x3
    g = torch.Generator().manual_seed(Ci + H + Co)
    x = torch.randn((N, Ci, H, W), generator=g)
    w = torch.randn((Co, Ci, 3, 3), generator=g) / (Ci * 9) ** 0.5
    b = torch.randn((Co,), generator=g) if bias else None
    gy = torch.randn((N, Co, H, W), generator=g)
    xr = x.double().requires_grad_(True); wr = w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    ref = F.conv2d(xr, wr, br, padding=1)
    ref.backward(gy.double())
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True) if bias else None
    assert eligible(xd, wd)
    y = skinny_conv3x3(xd, wd, bd)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gy.to(DEV))

    def close(a, r, tol, what):
        a = a.detach().double().cpu(); r = r.detach()
        scale = float(r.abs().max()) + 1e-12
        assert float((a - r).abs().max()) <= tol * scale, (what, float((a - r).abs().max()), scale)
    close(y, ref, 5e-6, "y")
    close(xd.grad, xr.grad, 5e-6, "grad_x")
    close(wd.grad, wr.grad, 2e-5, "grad_weight")          # reduction over N*H*W pixels in fp32 partials
    if bias:
        close(bd.grad, br.grad, 2e-5, "grad_bias")


def test_module_surgery_and_reproducibility():
    from distill_bev_amd.skinny_conv import SkinnyConv2d, use_skinny_convs
    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(64, 64, 3, padding=1), nn.ReLU(), nn.Conv2d(64, 2, 3, padding=1),
                        nn.Conv2d(2, 2, 1)).to(DEV).to(memory_format=torch.channels_last)
    keys = list(net.state_dict().keys())
    x = torch.randn((2, 64, 32, 32), device=DEV).contiguous(memory_format=torch.channels_last)
    ref = net(x)
    assert use_skinny_convs(net) == 1 and isinstance(net[2], SkinnyConv2d) and use_skinny_convs(net) == 0
    assert list(net.state_dict().keys()) == keys
    out = net(x)
    assert float((out - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    outs = []
    hid = net[1](net[0](x)).detach().requires_grad_(True)    # (MIOpen's own weight/data-gradient kernels use atomics)
    for _ in range(2):
        net.zero_grad(); hid.grad = None
        y = net[2](hid); y.square().sum().backward()
        outs.append((y.detach().clone(), hid.grad.clone(), net[2].weight.grad.clone(), net[2].bias.grad.clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))       # fixed-order reductions
    # NCHW input, CPU input: the module's own nn.Conv2d forward
    xn = x.contiguous()
    assert float((net[2](net[1](net[0](xn))) - net[2](net[1](net[0](x)))).abs().max()) < 1e-4
    cpu = nn.Sequential(nn.Conv2d(64, 2, 3, padding=1)); use_skinny_convs(cpu)
    assert cpu(torch.randn(1, 64, 4, 4)).shape == (1, 2, 4, 4)
