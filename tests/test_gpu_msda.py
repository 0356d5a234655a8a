"""GPU: multi-scale deformable attention kernels (csrc/msda.hip) through the reference's Function surface
(MultiScaleDeformableAttnFunction_fp32.apply) against the fp64 oracle (oracle/msda.py): forward 1e-5 of the output scale,
the three gradients 2e-5 of theirs; samples outside the maps; BEVFormer's shapes (8 heads x 32 channels, 4 levels, 8 points;
the temporal self-attention's single 50x50 level); run-to-run bit identity of forward and backward."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(seed, B, Q, NH, D, shapes, P):
    g = torch.Generator().manual_seed(seed)
    S = sum(h * w for h, w in shapes)
    value = torch.randn((B, S, NH, D), generator=g)
    loc = torch.rand((B, Q, NH, len(shapes), P, 2), generator=g) * 1.3 - 0.15
    att = torch.softmax(torch.randn((B, Q, NH, len(shapes) * P), generator=g), -1).view(B, Q, NH, len(shapes), P)
    gout = torch.randn((B, Q, NH * D), generator=g)
    return value, list(shapes), loc, att, gout


@pytest.mark.parametrize("B,Q,NH,D,shapes,P", [
    (2, 37, 2, 4, ((5, 7), (3, 4)), 3),                              # D/4 = 1 lane per row, ragged sizes
    (1, 50, 8, 32, ((20, 30), (10, 15), (5, 8), (3, 4)), 8),         # BEVFormer spatial cross-attention geometry
    (2, 64, 8, 32, ((50, 50),), 4),                                  # temporal self-attention: one BEV level
    (1, 20, 4, 64, ((6, 6), (3, 3)), 5),                             # 16 lanes per row, P not a multiple of the batch
])
def test_msda_forward_backward_vs_fp64_oracle(B, Q, NH, D, shapes, P):
    from distill_bev_amd.msda import MultiScaleDeformableAttnFunction_fp32 as F32
    from oracle import msda as OM
    dev = torch.device("cuda:0")
    value, shapes, loc, att, gout = _case(7 + D, B, Q, NH, D, shapes, P)
    v64, l64, a64 = [t.double().requires_grad_(True) for t in (value, loc, att)]
    ref = OM.msda_grid_sample(v64, shapes, l64, a64)
    rg = torch.autograd.grad(ref, (v64, l64, a64), gout.double())
    vd, ld, ad = [t.to(dev).requires_grad_(True) for t in (value, loc, att)]
    ss = torch.tensor(shapes, dtype=torch.long, device=dev)
    st = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
    out = F32.apply(vd, ss, st, ld, ad, 64)
    assert out.shape == (B, Q, NH * D)
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) <= 1e-5 * float(ref.abs().max())
    gv, gl, ga = torch.autograd.grad(out, (vd, ld, ad), gout.to(dev))
    for name, got, want in (("value", gv, rg[0]), ("loc", gl, rg[1]), ("attn", ga, rg[2])):
        err = float((got.cpu().double() - want).abs().max())
        assert err <= 2e-5 * float(want.abs().max()), (name, err, float(want.abs().max()))
    # bit reproducibility (no float atomics anywhere)
    out2 = F32.apply(vd, ss, st, ld, ad, 64)
    gv2, gl2, ga2 = torch.autograd.grad(out2, (vd, ld, ad), gout.to(dev))
    assert torch.equal(out, out2) and torch.equal(gv, gv2) and torch.equal(gl, gl2) and torch.equal(ga, ga2)


@pytest.mark.parametrize("mode", ["bevformer_sca_full_size", "hot_cells"])
def test_msda_full_size_and_heavily_shared_cells(mode):
    """(a) the spatial cross-attention geometry of the shipped BEVFormer recipe at full size -- 6 cameras, 4 FPN levels of a
    928 x 1600 image (116x200 ... 15x25 = 30 825 keys), 8 heads x 32 channels, 8 points -- against the fp64 oracle on 3 000
    queries per camera; (b) every sample of 4 000 queries inside the same 3 x 3 cells of one level: value rows shared by ~10^5
    samples (the long-segment path of the gradient's per-bin sort).  Forward 1e-5, gradients 2e-5 (value gradient of (b):
    1e-4 of its scale, fp32 sums of 10^5 terms vs fp64), bit-identical when repeated."""
    from distill_bev_amd.msda import MultiScaleDeformableAttnFunction_fp32 as F32
    from oracle import msda as OM
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    if mode == "bevformer_sca_full_size":
        B, Q, NH, D, P, shapes, vtol = 6, 3000, 8, 32, 8, [(116, 200), (58, 100), (29, 50), (15, 25)], 2e-5
        loc = torch.rand((B, Q, NH, 4, P, 2), generator=g) * 1.1 - 0.05
    else:
        B, Q, NH, D, P, shapes, vtol = 1, 4000, 8, 32, 8, [(40, 40), (20, 20)], 1e-4
        loc = 0.5 + (torch.rand((B, Q, NH, 2, P, 2), generator=g) - 0.5) * 0.06
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    value = torch.randn((B, S, NH, D), generator=g)
    att = torch.softmax(torch.randn((B, Q, NH, L * P), generator=g), -1).view(B, Q, NH, L, P)
    gout = torch.randn((B, Q, NH * D), generator=g)
    v64, l64, a64 = [t.double().requires_grad_(True) for t in (value, loc, att)]
    ref = OM.msda_grid_sample(v64, shapes, l64, a64)
    rg = torch.autograd.grad(ref, (v64, l64, a64), gout.double())
    vd, ld, ad = [t.to(dev).requires_grad_(True) for t in (value, loc, att)]
    ss = torch.tensor(shapes, dtype=torch.long, device=dev)
    st = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
    out = F32.apply(vd, ss, st, ld, ad, 64)
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) <= 1e-5 * float(ref.abs().max())
    gv, gl, ga = torch.autograd.grad(out, (vd, ld, ad), gout.to(dev))
    # the location gradient of bilinear sampling jumps where a sample crosses a pixel centre: the few of the 10^6 samples that
    # sit within fp32 rounding of one (|frac| < 1e-4 pixel) may legitimately fall on the other side than the fp64 oracle's
    wh = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float64).view(1, 1, 1, L, 1, 2)
    pix = loc.double() * wh - 0.5
    smooth = ((pix - pix.round()).abs() > 1e-4).all(-1, keepdim=True).expand_as(loc)
    assert float(smooth.double().mean()) > 0.999
    for name, got, want, tol in (("value", gv, rg[0], vtol), ("loc", gl, rg[1], 2e-5), ("attn", ga, rg[2], 2e-5)):
        d = (got.cpu().double() - want).abs()
        if name == "loc":
            d = d * smooth
        assert float(d.max()) <= tol * float(want.abs().max()), (name, float(d.max()), float(want.abs().max()))
    out2 = F32.apply(vd, ss, st, ld, ad, 64)
    gv2, gl2, ga2 = torch.autograd.grad(out2, (vd, ld, ad), gout.to(dev))
    assert torch.equal(out, out2) and torch.equal(gv, gv2) and torch.equal(gl, gl2) and torch.equal(ga, ga2)
    # linearity in value (a size-independent property of the op)
    v2 = torch.randn(value.shape, generator=g).to(dev)
    both = F32.apply(vd.detach() + v2, ss, st, ld.detach(), ad.detach(), 64)
    other = F32.apply(v2, ss, st, ld.detach(), ad.detach(), 64)
    assert float((both - out.detach() - other).abs().max()) <= 1e-5 * float(both.abs().max())


def test_msda_fp16_surface_and_refusal_of_cpu_tensors():
    from distill_bev_amd import _lib
    from distill_bev_amd.msda import MultiScaleDeformableAttnFunction_fp16 as F16, multi_scale_deformable_attn
    dev = torch.device("cuda:0")
    value, shapes, loc, att, gout = _case(3, 1, 16, 8, 32, ((8, 8),), 4)
    ss = torch.tensor(shapes, dtype=torch.long)
    st = torch.zeros(1, dtype=torch.long)
    with pytest.raises(_lib.DbevHipError):
        multi_scale_deformable_attn(value, ss, st, loc, att)
    vh = value.to(dev).half().requires_grad_(True)
    out = F16.apply(vh, ss.to(dev), st.to(dev), loc.to(dev).half(), att.to(dev).half(), 64)
    assert out.dtype == torch.float16
    ref = multi_scale_deformable_attn(value.to(dev).half().float(), ss.to(dev), st.to(dev), loc.to(dev).half().float(),
                                      att.to(dev).half().float())
    assert float((out.float() - ref).abs().max()) < 2e-3 * float(ref.abs().max())
    (g,) = torch.autograd.grad(out, vh, gout.to(dev).half())
    assert g.dtype == torch.float16 and bool(torch.isfinite(g).all())
