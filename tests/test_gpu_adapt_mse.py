"""GPU: the fused 1x1-adaptation + masked-MSE MFMA kernel (csrc/adapt_mse.hip) against
  (a) the unfused product sequence it replaces (MIOpen 1x1 conv -> abs-mean kernel -> masked-MSE kernels) and
  (b) the fp64 oracle (oracle/distill.py, pinned against the imported bevdet_distill.py by tests/golden/fgd_losses.npz),
losses 1e-4 relative, gradients of the student input / conv weight / bias 2e-4 of their scale.  Shapes cover every tile
variant (Ct = 32, 64, 96, 128 per slice; 6 slices of 64 channels at Ct = 384), a pixel count that is not a multiple of the 128-pixel
tile, the fp term on / off, and the recipe's real channel counts (256 -> 384)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _case(B, Cs, Ct, H, W, seed, use_fp):
    g = torch.Generator().manual_seed(seed)
    dev = torch.device("cuda:0")
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    x = cl(torch.randn((B, Cs, H, W), generator=g))
    teacher = cl(torch.randn((B, Ct, H, W), generator=g) * (1 + torch.rand((1, Ct, 1, 1), generator=g)))
    conv = nn.Conv2d(Cs, Ct, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn((Ct, Cs, 1, 1), generator=g) / Cs ** 0.5)
        conv.bias.copy_(torch.randn((Ct,), generator=g) * 0.1)
    conv = conv.to(dev).to(memory_format=torch.channels_last)
    fg = (torch.rand((B, 1, H, W), generator=g) < 0.1).float().to(dev)
    fg_scale = (fg * torch.rand((B, 1, H, W), generator=g).to(dev)).contiguous()
    n_bg = H * W - fg.sum((1, 2, 3))
    bg_scale = (1.0 / n_bg).view(B, 1, 1, 1).expand(B, 1, H, W).contiguous()
    fp = fp_scale = n_fp = None
    if use_fp:
        fp = ((torch.rand((B, 1, H, W), generator=g).to(dev) < 0.05) & (fg == 0)).float()
        n_fp = fp.sum((1, 2, 3))
        fp_scale = fp / n_fp.clamp(min=1).view(B, 1, 1, 1)
    return x, teacher, conv, fg, fg_scale, bg_scale, fp, fp_scale, n_fp


@pytest.mark.parametrize("B,Cs,Ct,H,W,use_fp", [(2, 64, 128, 16, 16, True), (2, 32, 96, 16, 16, False), (1, 64, 64, 10, 10, True),
                                                  (3, 96, 32, 12, 20, True), (2, 256, 384, 32, 32, True)])
def test_fused_adapt_mse_matches_unfused_sequence_and_fp64_oracle(B, Cs, Ct, H, W, use_fp):
    from distill_bev_amd.distill_loss import fgd_feature_losses, fgd_feature_losses_fused_adapt, fused_adapt_eligible
    from oracle import distill as OD
    x, teacher, conv, fg, fg_scale, bg_scale, fp, fp_scale, n_fp = _case(B, Cs, Ct, H, W, 5 + Ct, use_fp)
    assert fused_adapt_eligible(conv, x, teacher)
    kw = dict(w_fg=6e-3, w_bg=4e-2, fp=fp, fp_scale=fp_scale, n_fp=n_fp, w_fp=6e-2)
    res = {}
    for name in ("fused", "unfused"):
        xi = x.clone().requires_grad_(True)
        conv.zero_grad(set_to_none=True)
        if name == "fused":
            losses, att, c_att, pools = fgd_feature_losses_fused_adapt(xi, conv, teacher, fg, fg_scale, bg_scale, **kw)
        else:
            losses, att, c_att, pools = fgd_feature_losses(conv(xi), teacher, fg, fg_scale, bg_scale, **kw)
        total = sum(losses.values()) + 0.3 * (pools[1] * torch.linspace(-1, 1, H * W, device=x.device).view(1, 1, H, W)).sum()
        total.backward()
        res[name] = dict(losses={k: float(v) for k, v in losses.items()}, gx=xi.grad.clone(), gw=conv.weight.grad.clone(),
                         gb=conv.bias.grad.clone(), att=att.clone(), pool=pools[1].detach().clone())
    f, u = res["fused"], res["unfused"]
    for k in u["losses"]:
        assert abs(f["losses"][k] - u["losses"][k]) <= 1e-4 * abs(u["losses"][k]), (k, f["losses"][k], u["losses"][k])
    assert float((f["att"] - u["att"]).abs().max()) <= 1e-4 * float(u["att"].abs().max())
    assert float((f["pool"] - u["pool"]).abs().max()) <= 1e-5 * max(1.0, float(u["pool"].abs().max()))
    for k in ("gx", "gw", "gb"):
        assert float((f[k] - u[k]).abs().max()) <= 2e-4 * float(u[k].abs().max()), k
    # fp64 yardstick (oracle)
    S = OD.conv1x1(x.cpu().numpy(), conv.weight.detach().cpu().numpy(), conv.bias.detach().cpu().numpy())
    o, _ = OD.fgd_feature_losses(S, teacher.cpu().numpy(), fg.cpu().numpy(), fg_scale.cpu().numpy(), bg_scale.cpu().numpy(),
                                 fp=None if fp is None else fp.cpu().numpy().astype(np.float64),
                                 fp_scale=None if fp is None else fp_scale.cpu().numpy().astype(np.float64),
                                 n_fp=None if fp is None else n_fp.cpu().numpy().astype(np.float64))
    for k in o:
        assert abs(f["losses"][k] - o[k]) <= 1e-4 * abs(o[k]), (k, f["losses"][k], o[k])


def test_fused_adapt_mse_is_bit_reproducible_and_used_by_the_detector():
    from distill_bev_amd.distill_loss import _FusedAdaptMSE
    x, teacher, conv, *_ = _case(2, 256, 384, 32, 32, 1, True)
    cc = torch.rand((2, 384), device=x.device)
    a = _FusedAdaptMSE.apply(x, conv.weight, conv.bias, teacher, cc)
    b = _FusedAdaptMSE.apply(x, conv.weight, conv.bias, teacher, cc)
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    # the reference-fixture test of the detector's fgd_distill_loss (head position: 12 -> 16 channels) is NOT eligible
    # (channel counts below the MFMA tile); the recipe's 256 -> 384 head is:
    from distill_bev_amd.distill_loss import fused_adapt_eligible
    assert fused_adapt_eligible(conv, x, teacher)
    assert not fused_adapt_eligible(nn.Conv2d(12, 16, 1).to(x.device), x[:, :12].contiguous(memory_format=torch.channels_last),
                                    teacher[:, :16].contiguous(memory_format=torch.channels_last))
