"""TEST INFRASTRUCTURE -- CPU/torch restatement of DCNv2 (mmcv-full 1.6.0 modulated_deform_conv, an
un-vendored dependency of the reference: "parity unpinned" by reference fixtures).

Pinned instead by (i) a naive per-pixel loop implementation of the published definition
(tests/test_config_and_model.py) and (ii) equality with an ordinary convolution at zero offsets / unit mask.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(distill_bev_amd/dcn.py) runs the HIP kernels and has no torch fallback.
"""
import torch
import torch.nn.functional as F


def modulated_deform_conv2d(x, offset, mask, weight, bias, stride=1, padding=1, dilation=1):
    """DCNv2 forward (deform_groups=1, groups=1) restated with one grid_sample:
    out[n,o,h,w] = sum_{c,k} W[o,c,k] * mask[n,k,h,w] * bilinear(x[n,c], p_k + offset_k), zero padding.
    offset channels: (dy_0, dx_0, dy_1, dx_1, ...) for the kh*kw taps in row-major order (mmcv
    modulated_deform_conv CUDA kernel convention)."""
    N, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    K = kh * kw
    Ho = (H + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    dev, dt = x.device, x.dtype
    ys = torch.arange(Ho, device=dev, dtype=dt) * stride - padding
    xs = torch.arange(Wo, device=dev, dtype=dt) * stride - padding
    ky = (torch.arange(kh, device=dev, dtype=dt) * dilation).repeat_interleave(kw)
    kx = (torch.arange(kw, device=dev, dtype=dt) * dilation).repeat(kh)
    off = offset.view(N, K, 2, Ho, Wo)
    py = ys.view(1, 1, Ho, 1) + ky.view(1, K, 1, 1) + off[:, :, 0]
    px = xs.view(1, 1, 1, Wo) + kx.view(1, K, 1, 1) + off[:, :, 1]
    gx = 2.0 * px / max(W - 1, 1) - 1.0
    gy = 2.0 * py / max(H - 1, 1) - 1.0
    grid = torch.stack((gx, gy), -1).view(N, K * Ho, Wo, 2)
    cols = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    cols = cols.view(N, C, K, Ho, Wo) * mask.view(N, 1, K, Ho, Wo)
    # the contraction over (c, k) is a 1x1 convolution of the sampled columns: MIOpen's 1x1 conv
    # kernels run it ~40x faster than the skinny hipBLASLt GEMM torch.einsum lowers to (measured:
    # 21 ms -> 0.5 ms per call at N=48, C*K=2304, Ho*Wo=704)
    return F.conv2d(cols.reshape(N, C * K, Ho, Wo), weight.reshape(Co, C * K, 1, 1), bias)


def pack_forward(mod, x):
    """mmcv ModulatedDeformConv2dPack.forward with the op above (mod: distill_bev_amd.dcn.ModulatedDeformConv2dPack)."""
    out = mod.conv_offset(x)
    o1, o2, mask = torch.chunk(out, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)
    return modulated_deform_conv2d(x, offset, torch.sigmoid(mask), mod.weight, mod.bias, mod.stride, mod.padding,
                                   mod.dilation)
