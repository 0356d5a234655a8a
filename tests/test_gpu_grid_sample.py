"""GPU: channels-last bilinear grid_sample forward (csrc/upsample.hip dbev_grid_sample_bilinear_nhwc) through BEVDepth4D.shift_feature's
call against ATen's F.grid_sample(bilinear, zeros, align_corners=True): same source index and corner weights (rounded exactly as ATen
rounds them), the four-term sum within a few ulp (1e-6 of scale: the two builds contract its multiply-adds differently), including
sample points outside the map; gradients requested -> ATen's op (autograd) is used."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("N,C,H,W", [(2, 80, 128, 128), (1, 8, 7, 9), (3, 64, 16, 20)])
def test_grid_sample_kernel_matches_aten(N, C, H, W):
    from distill_bev_amd import _lib as L
    g = torch.Generator().manual_seed(N + C)
    x = torch.randn((N, C, H, W), generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    grid = (torch.rand((N, H, W, 2), generator=g) * 2.6 - 1.3).to(DEV)          # ~20 % of the points outside [-1, 1]
    grid[0, 0, 0] = torch.tensor([-1.0, 1.0]); grid[0, 0, 1] = torch.tensor([1.0, -1.0])   # exact corners
    ref = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    out = torch.empty_like(x)
    L.call("dbev_grid_sample_bilinear_nhwc", L.ptr(x), L.ptr(grid.contiguous()), N, C, H, W, H, W, L.ptr(out), L.stream_ptr(x.device))
    assert out.is_contiguous(memory_format=torch.channels_last)
    assert float((out - ref).abs().max()) <= 1e-6 * float(ref.abs().max()), float((out - ref).abs().max())
    assert float((ref == 0).float().mean()) > 0.01                                  # the zero-padding branch was exercised


def test_shift_feature_takes_the_kernel_without_grad_and_atens_op_with_grad():
    from distill_bev_amd.detectors import BEVDepth4DDistill
    import types
    vt = types.SimpleNamespace(dx=torch.tensor([0.8, 0.8, 20.0], device=DEV), bx=torch.tensor([-50.8, -50.8, 0.0], device=DEV))
    me = types.SimpleNamespace(img_view_transformer=vt, interpolation_mode="bilinear", __dict__={})
    me._feat2bev = types.MethodType(BEVDepth4DDistill._feat2bev, me)
    torch.manual_seed(0)
    n, c, h, w = 2, 16, 32, 32
    x = torch.randn((n, c, h, w), device=DEV).contiguous(memory_format=torch.channels_last)
    rots = [torch.eye(3, device=DEV).expand(n, 6, 3, 3).contiguous() for _ in range(2)]
    trans = [torch.zeros((n, 6, 3), device=DEV), torch.zeros((n, 6, 3), device=DEV)]
    trans[1][:, :, 0] = 1.7; trans[1][:, :, 1] = -0.9                           # ego motion between the frames
    a = BEVDepth4DDistill.shift_feature(me, x, trans, rots)
    xg = x.clone().requires_grad_(True)
    b = BEVDepth4DDistill.shift_feature(me, xg, trans, rots)
    assert a.grad_fn is None and b.grad_fn is not None
    assert float((a - b.detach()).abs().max()) <= 1e-6 * float(b.detach().abs().max())
