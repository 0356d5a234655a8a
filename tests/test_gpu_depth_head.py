"""GPU: fused depth head (csrc/depth_head.hip, distill_bev_amd/depth_head.py) through the C ABI against the reference's module
sequence nn.BatchNorm2d -> nn.Conv2d(c, D, 1) -> softmax(dim=1) (view_transformer_mine.py:300-309, 325-328) evaluated in fp64 on
the host: depth_digit / depth_prob 1e-5 of scale, running statistics 1e-6, all five parameter / input gradients 2e-5 (both outputs
feed the loss, as in the step: the depth loss reads depth_digit, the lift reads depth_prob); eval mode; bit-identical repeats;
ineligible shapes take the module sequence and are counted."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _modules(C, D, seed):
    g = torch.Generator().manual_seed(seed)
    bn = nn.BatchNorm2d(C, eps=1e-5, momentum=0.1)
    conv = nn.Conv2d(C, D, 1)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g)); bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.2); conv.bias.copy_(torch.randn(D, generator=g))
    return bn, conv, g


def _close(a, r, tol, what):
    a, r = a.detach().double().cpu(), r.detach().double().cpu()
    err = float((a - r).abs().max()) / (float(r.abs().max()) + 1e-30)
    assert err < tol, f"{what}: {err:.3e}"


@pytest.mark.parametrize("BN,C,D,H,W", [(12, 256, 59, 16, 44), (3, 64, 59, 5, 7), (2, 128, 64, 9, 13), (1, 32, 7, 3, 5)])
def test_depth_head_training_forward_backward_vs_module_sequence(BN, C, D, H, W):
    from distill_bev_amd.depth_head import depth_head, eligible
    bn, conv, g = _modules(C, D, C + D)
    x = torch.randn((BN, C, H, W), generator=g) * 1.7 + 0.4
    gd = torch.randn((BN, D, H, W), generator=g, dtype=torch.float64)
    gp = torch.randn((BN, D, H, W), generator=g, dtype=torch.float64)
    # fp64 reference
    bn64, conv64 = nn.BatchNorm2d(C, eps=bn.eps, momentum=bn.momentum).double(), nn.Conv2d(C, D, 1).double()
    bn64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    conv64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    x64 = x.double().requires_grad_(True)
    digit64 = conv64(bn64.train()(x64))
    prob64 = digit64.softmax(dim=1)
    ((digit64 * gd).sum() + (prob64 * gp).sum()).backward()

    bn, conv = bn.to(DEV).train(), conv.to(DEV).to(memory_format=torch.channels_last)
    init = {k: v.clone() for k, v in bn.state_dict().items()}
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert eligible(xd, bn, conv)

    def run():
        bn.load_state_dict(init)
        digit, prob = depth_head(xd, bn, conv)
        assert type(digit.grad_fn).__name__ == "_DepthHeadBackward"
        loss = (digit * gd.float().to(DEV)).sum() + (prob * gp.float().to(DEV)).sum()
        return (digit, prob) + torch.autograd.grad(loss, [xd, bn.weight, bn.bias, conv.weight, conv.bias])
    out = run()
    assert out[0].is_contiguous(memory_format=torch.channels_last) and out[1].is_contiguous(memory_format=torch.channels_last)
    _close(out[0], digit64, 1e-5, "depth_digit")
    _close(out[1], prob64, 1e-5, "depth_prob")
    assert float((out[1].detach().sum(dim=1) - 1).abs().max()) < 1e-5
    _close(bn.running_mean, bn64.running_mean, 1e-6, "running_mean"); _close(bn.running_var, bn64.running_var, 2e-6, "running_var")
    assert int(bn.num_batches_tracked) == 1
    for got, want, what in zip(out[2:], (x64.grad, bn64.weight.grad, bn64.bias.grad, conv64.weight.grad, conv64.bias.grad),
                               ("dx", "dgamma", "dbeta", "dW", "db")):
        _close(got, want, 2e-5 * max(1.0, (BN * H * W) ** 0.5 / 16), what)
    out2 = run()
    assert all(torch.equal(a, b) for a, b in zip(out, out2))


def test_depth_head_eval_mode_no_grad_and_fallbacks():
    from distill_bev_amd import _lib as L
    from distill_bev_amd.depth_head import depth_head, eligible
    bn, conv, g = _modules(256, 59, 5)
    x = torch.randn((6, 256, 16, 44), generator=g)
    bn, conv = bn.to(DEV).eval(), conv.to(DEV)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        assert eligible(xd, bn, conv)
        digit, prob = depth_head(xd, bn, conv)
        ref = conv.double()(bn.double()(xd.double()))
        bn.float(); conv.float()
    _close(digit, ref, 1e-5, "eval depth_digit"); _close(prob, ref.softmax(dim=1), 1e-5, "eval depth_prob")
    # training mode under no_grad (the adjacent frame of BEVDepth4D): batch statistics, running statistics move, nothing is saved
    bn.train()
    before = bn.running_mean.clone()
    with torch.no_grad():
        digit2, prob2 = depth_head(xd, bn, conv)
        ref2 = conv(torch.nn.functional.batch_norm(xd, None, None, bn.weight, bn.bias, True, 0.0, bn.eps))
    _close(digit2, ref2, 1e-5, "train/no_grad depth_digit")
    assert not torch.equal(before, bn.running_mean) and digit2.grad_fn is None
    # not covered by the kernel: NCHW input, 96 depth bins -> module sequence, counted
    L._warned_fallbacks.clear(); L.fallback_reset()
    with pytest.warns(RuntimeWarning, match="depth_head"):
        d3, p3 = depth_head(x.to(DEV), bn, conv)
    wide = nn.Conv2d(256, 96, 1).to(DEV)
    d4, p4 = depth_head(xd, bn, wide)
    assert L.fallback_counts()["depth_head"] == 2 and p4.shape[1] == 96
    _close(d3, digit2, 1e-5, "fallback value")


def test_view_transformer_depth_branch_uses_the_fused_head_and_matches_the_module_sequence():
    """ViewTransformerLSSBEVDepth.depth_feat_and_prob: same depth_digit / depth_prob and the same gradient at the image feature as
    the module sequence (depth_head disabled), fallback ledger untouched."""
    from distill_bev_amd import _lib as L
    from distill_bev_amd import depth_head as DH
    import distill_bev_amd.nets  # noqa: F401  (registers ResNetForBEVDet)
    import distill_bev_amd.view_transformer  # noqa: F401
    from distill_bev_amd.registry import MODELS
    torch.manual_seed(0)
    cfg = dict(type="ViewTransformerLSSBEVDepth", loss_depth_weight=100.0,
               grid_config=dict(xbound=[-51.2, 51.2, 0.8], ybound=[-51.2, 51.2, 0.8], zbound=[-10.0, 10.0, 20.0], dbound=[1.0, 60.0, 1.0]),
               data_config=dict(input_size=(256, 704)), numC_input=64, numC_Trans=16,
               extra_depth_net=dict(type="ResNetForBEVDet", numC_input=32, num_layer=[1], num_channels=[32], stride=[1]))
    vt = MODELS.build(cfg).to(DEV).train()
    vt = vt.to(memory_format=torch.channels_last)
    B, N, fH, fW = 1, 6, 16, 44
    x = torch.randn((B * N, 64, fH, fW), device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    eye = torch.eye(3, device=DEV).expand(B, N, 3, 3).contiguous()
    z3 = torch.zeros((B, N, 3), device=DEV)
    state = {k: v.clone() for k, v in vt.state_dict().items()}
    res = []
    L.fallback_reset()
    for fused in (True, False):
        vt.load_state_dict(state)
        DH.set_enabled(fused)
        try:
            feat, digit, prob = vt.depth_feat_and_prob(x, eye, z3, eye, eye, z3)
            (gx,) = torch.autograd.grad((digit.sigmoid().sum() + (prob * prob).sum()), x)
        finally:
            DH.set_enabled(True)
        res.append((digit, prob, gx, type(digit.grad_fn).__name__))
    assert res[0][3] == "_DepthHeadBackward" and res[1][3] != "_DepthHeadBackward"
    assert L.fallback_counts()["depth_head"] == 1                       # the disabled pass
    for a, b, what in zip(res[0][:3], res[1][:3], ("depth_digit", "depth_prob", "grad at the image feature")):
        _close(a, b, 2e-5, what)
