"""GPU: the fp32-MFMA 1x1 convolution with BatchNorm statistics in its epilogue (csrc/conv1x1.hip, dbev_conv1x1_forward) and its
use inside the bottleneck blocks (nets.Bottleneck._fused -> bn_act(..., pre=partial rows)).

The reference runs `conv1 -> norm1 -> act`, `conv3 -> norm3 (+ identity) -> act` as separate modules (bricks/res_block.py:102-230);
bars: convolution output 2e-6 of scale vs the fp64 definition, the partial statistics rows sum to the per-channel sums of the
output to 1e-6, bit-identical when repeated; a bottleneck block through the fused path equals the library path (F.conv2d + the
statistics pass) to 1e-5 in outputs, input / weight / norm gradients and running statistics."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,N", [(128 * 9 + 37, 64, 256), (5000, 32, 32), (4096, 64, 64), (3001, 256, 64), (2049, 128, 96),
                                   (70000, 64, 128), (1000, 512, 128), (777, 256, 512), (130, 1024, 256), (40000, 256, 128)])
def test_conv1x1_forward_and_partial_statistics(M, K, N):
    from distill_bev_amd import _lib as L
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn((M, K), generator=g).to(dev)
    w = (torch.randn((N, K), generator=g) / K ** 0.5).to(dev)
    rows = int(L.call("dbev_conv1x1_stats_rows", M, K, N))
    assert rows > 0
    outs = []
    for _ in range(2):
        y = torch.full((M, N), float("nan"), device=dev)
        part = torch.full((rows, 2, N), float("nan"), device=dev)
        L.call("dbev_conv1x1_forward", L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(part), M, K, N, K, L.stream_ptr(dev))
        outs.append((y, part))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    y, part = outs[0]
    ref = x.double() @ w.double().t()
    assert float((y.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    s = part.double().sum(0)
    assert float((s[0] - y.double().sum(0)).abs().max()) <= 1e-6 * float(y.double().abs().sum(0).max())
    assert float((s[1] - y.double().square().sum(0)).abs().max()) <= 1e-6 * float(y.double().square().sum(0).max())
    # without statistics: same output
    y2 = torch.empty_like(y)
    L.call("dbev_conv1x1_forward", L.ptr(x), L.ptr(w), L.ptr(y2), L.ptr(None), M, K, N, K, L.stream_ptr(dev))
    assert torch.equal(y, y2)
    # a row-strided input (channel slice of a wider tensor)
    xw = torch.randn((M, K + 32), generator=g).to(dev)
    L.call("dbev_conv1x1_forward", L.ptr(xw), L.ptr(w), L.ptr(y2), L.ptr(None), M, K, N, K + 32, L.stream_ptr(dev))
    ref2 = xw[:, :K].double() @ w.double().t()
    assert float((y2.double() - ref2).abs().max()) <= 2e-6 * float(ref2.abs().max())


def test_conv1x1_rejects_unsupported_shapes():
    from distill_bev_amd import _lib as L
    assert int(L.call("dbev_conv1x1_stats_rows", 1000, 48, 64)) == 0        # Cin % 32
    assert int(L.call("dbev_conv1x1_stats_rows", 1000, 64, 24)) == 0        # Cout % 32
    assert int(L.call("dbev_conv1x1_stats_rows", 0, 64, 64)) == 0


@pytest.mark.parametrize("inplanes,planes,with_ds", [(64, 64, True), (256, 64, False), (128, 32, False)])
def test_bottleneck_through_the_fused_gemm_equals_the_library_path(inplanes, planes, with_ds):
    from distill_bev_amd import bn_act as BA
    from distill_bev_amd import _lib as L
    from distill_bev_amd.nets import Bottleneck
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4)) if with_ds else None
    blk = Bottleneck(inplanes, planes, downsample=ds).to(dev).to(memory_format=torch.channels_last).train()
    for m in blk.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    x0 = torch.randn(4, inplanes, 24, 40, device=dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(4, planes * 4, 24, 40, device=dev).contiguous(memory_format=torch.channels_last)
    sd0 = {k: v.clone() for k, v in blk.state_dict().items()}
    res = {}
    old = dict(BA._C1)
    try:
        for mode in ("library", "fused"):
            BA._C1.update(enabled=mode == "fused", min_rows=1)
            for warm in (True, False):       # first pass: the convolution library's first-touch work for these shapes, discarded
                blk.load_state_dict(sd0)
                L.kernel_timing(["c1x1_fwd"]); L.kernel_timing_read()
                x = x0.clone().requires_grad_(True)
                y = blk(x)
                if warm:
                    torch.autograd.grad(y, [x] + list(blk.parameters()), gy)
            n_launch = len(L.kernel_timing_read().get("c1x1_fwd", []))
            L.kernel_timing(False)
            assert n_launch == ((3 if with_ds else 2) if mode == "fused" else 0), (mode, n_launch)
            params = dict(blk.named_parameters())
            grads = torch.autograd.grad(y, [x] + list(params.values()), gy)
            res[mode] = dict(y=y.detach(), gx=grads[0], **{"g_" + n: gr for n, gr in zip(params, grads[1:])},
                             **{"s_" + n: b.clone() for n, b in blk.named_buffers() if "running" in n})
    finally:
        BA._C1.clear(); BA._C1.update(old)
    for k in res["library"]:
        a, b = res["library"][k].double(), res["fused"][k].double()
        err = float((a - b).norm() / a.norm().clamp_min(1e-12))
        assert err <= 1e-5, (k, err)
