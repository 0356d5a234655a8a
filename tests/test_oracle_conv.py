"""CPU: the convolution oracle (oracle/conv.py) against the op the reference calls (torch.nn.functional.conv2d), the Winograd identity
the HIP kernels evaluate, and the host logic of the Winograd / GEMM module swaps (no GPU: the stock convolution runs)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def test_direct_and_winograd_restatements_equal_torch_conv2d():
    from oracle import conv as OC
    rng = np.random.default_rng(0)
    x = rng.normal(size=(2, 5, 6, 8)); w = rng.normal(size=(7, 5, 3, 3)); b = rng.normal(size=(7,))
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), 1, 1).numpy()
    assert np.abs(OC.conv2d_direct(x, w, b) - ref).max() < 1e-12
    assert np.abs(OC.conv3x3_winograd(x, w) + b[None, :, None, None] - ref).max() < 1e-12
    w1 = rng.normal(size=(4, 5, 1, 1))
    assert np.abs(OC.conv2d_direct(x, w1) - F.conv2d(torch.from_numpy(x), torch.from_numpy(w1)).numpy()).max() < 1e-12


def test_module_swaps_keep_keys_and_fall_back_to_the_stock_convolution_without_a_device_tensor(monkeypatch):
    from distill_bev_amd import gemm1x1, wino
    net = nn.Sequential(nn.Conv2d(64, 64, 3, padding=1, bias=False), nn.Conv2d(64, 128, 1, bias=False), nn.Conv2d(64, 3, 3, padding=1),
                        nn.Conv2d(64, 64, 3, stride=2, padding=1))
    keys = list(net.state_dict())
    assert wino.use_wino_convs(net) == 1 and type(net[0]) is wino.WinoConv2d and type(net[2]) is nn.Conv2d and type(net[3]) is nn.Conv2d
    assert gemm1x1.use_gemm_convs(net) == 0                        # opt-in
    monkeypatch.setattr(gemm1x1, "_ON", True)
    assert gemm1x1.use_gemm_convs(net) == 1 and type(net[1]) is gemm1x1.GemmConv2d
    assert list(net.state_dict()) == keys
    x = torch.randn(1, 64, 8, 8)
    assert not wino.eligible(x, net[0].weight) and not gemm1x1.eligible(x, net[1].weight)
    assert torch.equal(net[0](x), F.conv2d(x, net[0].weight, None, 1, 1)) and torch.equal(net[1](x), F.conv2d(x, net[1].weight))
    # the work-item rule: 64-tile blocks x 64-channel blocks
    assert wino._blocks(48, 16, 44) == 48 * 3 and wino._blocks(8, 128, 128) == 8 * 64
