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
