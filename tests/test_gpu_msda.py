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
