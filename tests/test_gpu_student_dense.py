"""GPU: the student's dense modules (nets.py) against the reference's own resnet.py / res_block.py / lss_fpn.py / fpn.py
(tests/golden/student_dense.npz) in the three configurations the product runs them in: NCHW (MIOpen + torch norms), channels-last
with the fused norm-act kernels (bn_act / bn_act_dual / BatchNormAct2d -- the bench configuration), and channels-last with the
HIP bilinear upsampling swapped in (accelerate_modules)."""
import os

import numpy as np
import pytest
import torch

from test_student_dense import GOLD, cases, run_case

pytestmark = pytest.mark.gpu


def _accelerate(m):
    from distill_bev_amd.train_step import accelerate_modules
    n_bn, n_up = accelerate_modules(m)
    return n_bn, n_up


@pytest.mark.parametrize("mode", ["nchw", "channels_last_fused"])
@pytest.mark.parametrize("tag", list(cases()))
def test_student_dense_modules_match_the_reference_files(tag, mode):
    fx = np.load(os.path.join(GOLD, "student_dense.npz"))
    dev = torch.device("cuda:0")
    cl = mode != "nchw"
    err = run_case(fx, tag, cases()[tag], dev, channels_last=cl, prepare=_accelerate if cl else None)
    print(tag, mode, err)
    # outputs / statistics 2e-5; gradients through 6-12 training-mode norms 1e-4 (CPU vs GPU convolution rounding)
    assert err["y"] <= 2e-5 and err["eval"] <= 2e-5 and err["stats"] <= 2e-5, err
    assert err["gx"] <= 1e-4 and err["gp"] <= 1e-4, err


def test_resnet101_stage_plan_builds_and_runs():
    """configs[4] names a ResNet-101 student: the 101 arch entry (3, 4, 23, 3 bottlenecks) builds, carries mmdet's key set and
    runs forward + backward on the fused kernels"""
    from distill_bev_amd import nets
    dev = torch.device("cuda:0")
    m = nets.ResNet(depth=101, out_indices=(2, 3), norm_eval=False, zero_init_residual=False).to(dev).to(memory_format=torch.channels_last)
    keys = set(m.state_dict())
    assert "layer3.22.conv3.weight" in keys and "layer4.2.bn3.running_var" in keys and "layer1.0.downsample.1.weight" in keys
    assert sum(p.numel() for p in m.parameters()) == 42_500_160          # torchvision resnet101 minus the fc layer
    x = torch.randn(2, 3, 64, 96, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    c4, c5 = m.train()(x)
    assert c4.shape == (2, 1024, 4, 6) and c5.shape == (2, 2048, 2, 3)
    (c4.square().mean() + c5.square().mean()).backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0


@pytest.mark.parametrize("N,C,H,W", [(3, 64, 17, 22), (2, 8, 9, 9), (1, 64, 128, 352), (2, 4, 2, 3)])
def test_stem_max_pool_kernels_vs_aten_values_ties_and_gradients(N, C, H, W):
    """csrc/maxpool.hip (MaxPool2d(3, 2, 1), channels-last) vs ATen: output bit-equal, input gradient bit-equal -- including the tied
    zeros a ReLU leaves (the FIRST maximum of a window in scan order wins in both) and windows with NaN."""
    import torch.nn as nn
    from distill_bev_amd.pool import max_pool, _MaxPool3x3s2
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N * 100 + W)
    x = torch.relu(torch.randn((N, C, H, W), generator=g))                  # ~half of the values are exactly 0: ties everywhere
    x = (x * 4).round() / 4                                                 # and ties among the positive values
    if H * W > 50:
        x[0, 1, 3, 4] = float("nan")
    gy_seed = torch.randn((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), generator=g)
    pool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
    xa = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xb = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ya = max_pool(pool, xa)
    assert type(ya.grad_fn).__name__ == "_MaxPool3x3s2Backward" and ya.is_contiguous(memory_format=torch.channels_last)
    yb = pool(xb)
    gy = gy_seed.to(dev).contiguous(memory_format=torch.channels_last)
    (ga,) = torch.autograd.grad(ya, xa, gy)
    (gb,) = torch.autograd.grad(yb, xb, gy)
    assert torch.equal(torch.nan_to_num(ya, nan=-7.0), torch.nan_to_num(yb, nan=-7.0))
    assert torch.equal(ga, gb), float((ga - gb).abs().max())
    # other geometries / layouts keep the module
    assert max_pool(nn.MaxPool2d(2, 2), xa).grad_fn is not None and type(max_pool(nn.MaxPool2d(2, 2), xa).grad_fn).__name__ != "_MaxPool3x3s2Backward"
    assert type(max_pool(pool, x.to(dev).requires_grad_(True)).grad_fn).__name__ != "_MaxPool3x3s2Backward" or C == 1


@pytest.mark.parametrize("N,C,H,W", [(3, 64, 18, 22), (2, 64, 64, 88), (1, 8, 9, 9)])
def test_stem_norm_relu_max_pool_in_one_pass_equals_the_three_module_sequence(N, C, H, W):
    """round 5: pool.norm_relu_max_pool = mmdet ResNet's stem tail `norm1 -> relu -> maxpool` (training mode) as the fused norm's
    statistics + ONE normalise / rectify / pool pass: pooled output bit-equal to the product's own three-kernel sequence (bn_act +
    max_pool: the same coefficients, the same tie rule on the same values), gradients of the input and of gamma / beta equal to it,
    running statistics bit-equal; and against the plain torch modules within fp32 round-off."""
    import torch.nn as nn
    from distill_bev_amd import bn_act as BA
    from distill_bev_amd.pool import max_pool, norm_relu_max_pool
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N + W)
    x0 = (torch.randn((N, C, H, W), generator=g) * 2).to(dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    pool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
    res = []
    for which in ("fused", "sequence", "torch"):
        torch.manual_seed(5)
        bn = nn.BatchNorm2d(C).to(dev).train()
        with torch.no_grad():
            bn.weight.uniform_(-1.0, 1.5); bn.bias.normal_(0, 0.5)          # negative scales too: the ReLU must be applied per tap
        x = x0.clone().requires_grad_(True)
        if which == "fused":
            y = norm_relu_max_pool(bn, pool, x)
            assert type(y.grad_fn).__name__ == "_NormReluMaxPoolBackward"
        elif which == "sequence":
            y = max_pool(pool, BA.bn_act(x, bn, None, True))
        else:
            y = pool(torch.relu(bn(x)))
        gx, gw, gb = torch.autograd.grad(y, [x, bn.weight, bn.bias], gy)
        res.append((y.detach(), gx, gw, gb, bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked)))
    f, s, t = res
    assert torch.equal(f[0], s[0]) and torch.equal(f[4], s[4]) and torch.equal(f[5], s[5]) and f[6] == s[6] == t[6] == 1
    for a, b in zip(f[1:4], s[1:4]):       # (the fused backward gathers the pooled gradient inside the norm's two passes: its own summation order)
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * float(b.abs().max())), float((a - b).abs().max())
    assert torch.allclose(f[0], t[0], rtol=1e-5, atol=1e-5)
    for a, b in zip(f[1:4], t[1:4]):
        assert float((a - b).norm() / b.norm().clamp_min(1e-12)) < 1e-4
    assert torch.allclose(f[4], t[4], rtol=1e-6, atol=1e-6) and torch.allclose(f[5], t[5], rtol=1e-5, atol=1e-6)
    # the detached frame: no autograd, same values
    with torch.no_grad():
        torch.manual_seed(5)
        bn = nn.BatchNorm2d(C).to(dev).train()
        bn.weight.uniform_(-1.0, 1.5); bn.bias.normal_(0, 0.5)
        assert torch.equal(norm_relu_max_pool(bn, pool, x0), f[0])
    # eval-mode norms keep the module sequence
    assert type(norm_relu_max_pool(bn.eval(), pool, x0.clone().requires_grad_(True)).grad_fn).__name__ != "_NormReluMaxPoolBackward"
