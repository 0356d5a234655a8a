"""GPU parity: fg-mask rasteriser, |x| mean maps, fused masked-MSE (C ABI, HIP) vs the
reference fixtures (imported box_np_ops) and the fp64 oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from distill_bev_amd import synthetic as syn
from oracle import distill as OD

pytestmark = pytest.mark.gpu
CFG = dict(grid_size=[1024, 1024, 40], point_cloud_range=[-51.2, -51.2, -5.0, 51.2, 51.2, 3.0],
           voxel_size=[0.1, 0.1, 0.2])


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("H", [128, 64])
def test_fg_scale_mask_bit_exact_vs_reference_fixture(H):
    from distill_bev_amd.distill_loss import ForegroundMaskRasterizer
    dev = _dev()
    g = load_golden(f"fgmask_{H}.npz")
    r = ForegroundMaskRasterizer(**CFG)
    fg, fs, bs = r(H, H, [g["boxes0"], g["boxes1"], g["boxes2"]], dev)
    assert np.array_equal(fg.cpu().numpy(), g["fg"])
    # fg / bg_scale bit exact.  fg_scale: the reference value is torch.sqrt on the host CPU,
    # which is not correctly rounded and machine dependent (1 ulp) -> 1-ulp tolerance vs the
    # fixture, bit exact vs the (correctly rounded) oracle.
    assert np.allclose(fs.cpu().numpy(), g["fg_scale"], rtol=2.5e-7, atol=0)
    o_fg0, o_fs0, o_bs0 = OD.foreground_scale_mask(H, H, [g["boxes0"], g["boxes1"], g["boxes2"]])
    assert np.array_equal(fs.cpu().numpy(), o_fs0)
    assert np.array_equal(bs.cpu().numpy(), g["bg_scale"])
    # ragged / empty: a sample with no boxes, > 64 boxes in one sample (LDS tiling)
    many, _ = syn.gt_boxes(150, np.random.default_rng(0))
    fg2, fs2, bs2 = r(H, H, [np.zeros((0, 9), np.float32), many], dev)
    o_fg, o_fs, o_bs = OD.foreground_scale_mask(H, H, [np.zeros((0, 9), np.float32), many])
    assert np.array_equal(fg2.cpu().numpy(), o_fg) and float(fg2[0].sum()) == 0
    assert np.array_equal(fs2.cpu().numpy(), o_fs) and np.array_equal(bs2.cpu().numpy(), o_bs)


def test_abs_mean_maps():
    from distill_bev_amd.distill_loss import abs_mean_maps
    dev = _dev()
    for (B, C, H, W) in [(2, 384, 128, 128), (3, 5, 6, 10), (1, 130, 64, 64)]:
        x = torch.randn((B, C, H, W), device=dev)
        pix, ch = abs_mean_maps(x)
        assert torch.allclose(pix, x.abs().mean(1, keepdim=True), atol=1e-5, rtol=1e-5)
        assert torch.allclose(ch, x.abs().mean((2, 3), keepdim=True), atol=1e-5, rtol=1e-5)


def test_fgd_losses_vs_fp64_oracle_and_autograd():
    """head position shapes (B=2, C=384, 128x128) incl. the fp term; gradient vs the unfused
    torch expression."""
    from distill_bev_amd.distill_loss import ForegroundMaskRasterizer, fgd_feature_losses
    dev = _dev()
    rng = np.random.default_rng(5)
    B, C, H, W = 2, 384, 128, 128
    S = rng.normal(size=(B, C, H, W)).astype(np.float32)
    T = rng.normal(size=(B, C, H, W)).astype(np.float32)
    boxes = [syn.gt_boxes(30, rng)[0] for _ in range(B)]
    fg, fs, bs = OD.foreground_scale_mask(H, W, boxes)
    gt_hm = rng.uniform(0, 0.3, size=(B, 1, H, W)).astype(np.float32)
    t_hm = rng.uniform(0, 0.2, size=(B, 1, H, W)).astype(np.float32)
    fp, fpsc, nfp = OD.fp_masks(fg, gt_hm, t_hm, 0.1)
    ref, aux = OD.fgd_feature_losses(S, T, fg, fs, bs, fp, fpsc, nfp)
    r = ForegroundMaskRasterizer(**CFG)
    dfg, dfs, dbs = r(H, W, boxes, dev)
    St = torch.from_numpy(S).to(dev).requires_grad_(True)
    Tt = torch.from_numpy(T).to(dev)
    out, att, c_att, _ = fgd_feature_losses(
        St, Tt, dfg, dfs, dbs, w_fg=6e-3, w_bg=4e-2, w_fp=6e-2,
        fp=torch.from_numpy(fp.astype(np.float32)).to(dev),
        fp_scale=torch.from_numpy(fpsc.astype(np.float32)).to(dev),
        n_fp=torch.from_numpy(nfp.astype(np.float32)).to(dev))
    for k in ref:
        rel = abs(float(out[k]) - ref[k]) / max(abs(ref[k]), 1e-12)
        assert rel < 1e-4, (k, float(out[k]), ref[k])
    assert np.abs(att.cpu().numpy() - aux["att"]).max() < 1e-4 * aux["att"].max()
    total = out["kd_fg_feat_loss"] + out["kd_bg_feat_loss"] + 2.0 * out["kd_fp_bg_feat_loss"]
    total.backward()
    # unfused torch expression of the same losses (fp64 on device)
    S2 = torch.from_numpy(S).to(dev).double().requires_grad_(True)
    sq = (S2 - Tt.double()) ** 2
    fgw = torch.from_numpy(aux["fg_w"]).to(dev); bgw = torch.from_numpy(aux["bg_w"]).to(dev)
    fpw = torch.from_numpy(fp * fpsc * aux["att"] * aux["c_att"]).to(dev)
    t2 = (sq * fgw).sum() * 6e-3 / B + (sq * bgw).sum() * 4e-2 / B + 2.0 * (sq * fpw).sum() * 6e-2 / B
    t2.backward()
    gerr = (St.grad.double() - S2.grad).abs().max() / S2.grad.abs().max()
    assert float(gerr) < 1e-4


@pytest.mark.parametrize("B,C,H,W", [(2, 384, 128, 128), (3, 128, 17, 9), (1, 8, 5, 3), (2, 1024, 8, 8), (2, 260, 6, 7)])
def test_channels_last_loss_kernels_match_nchw_kernels_and_fp64(B, C, H, W):
    """NHWC variants (dbev_abs_mean_maps_nhwc, dbev_fgd_masked_mse_*_nhwc): same numbers as the NCHW kernels
    (summation order differs: 1e-6 relative) and as the fp64 torch expression; dS comes back channels-last."""
    from distill_bev_amd.distill_loss import _is_nhwc, abs_mean_maps, masked_mse_sums
    dev = _dev()
    g = torch.Generator().manual_seed(C + H)
    S = torch.randn((B, C, H, W), generator=g).to(dev); T = torch.randn((B, C, H, W), generator=g).to(dev)
    wf = torch.rand((B, 1, H, W), generator=g).to(dev); wb = torch.rand((B, 1, H, W), generator=g).to(dev)
    wp = torch.rand((B, 1, H, W), generator=g).to(dev); cc = torch.rand((B, C), generator=g).to(dev)
    Scl = S.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    Tcl = T.contiguous(memory_format=torch.channels_last)
    assert _is_nhwc(Scl) and not _is_nhwc(S)
    if (H * W) % 4:                       # the NCHW masked-MSE kernels need HW % 4 == 0; the NHWC ones do not
        out_c = masked_mse_sums(Scl, Tcl, wf, wb, wp, cc)
        sq = (S - T).double() ** 2
        ref = torch.stack([(sq * wf.double()).sum(), (sq * wb.double()).sum(),
                           (sq * wp.double() * cc.double().view(B, C, 1, 1)).sum()])
        assert float(((out_c.double() - ref).abs() / ref.abs()).max()) < 2e-6
        pix_c, ch_c = abs_mean_maps(Tcl)
        assert torch.allclose(pix_c, T.abs().mean(1, keepdim=True), atol=1e-6, rtol=1e-5)
        assert torch.allclose(ch_c, T.abs().mean((2, 3), keepdim=True), atol=1e-6, rtol=1e-5)
        return
    pix_n, ch_n = abs_mean_maps(T)
    pix_c, ch_c = abs_mean_maps(Tcl)
    assert torch.allclose(pix_c, T.abs().mean(1, keepdim=True), atol=1e-6, rtol=1e-5)
    assert torch.allclose(ch_c, T.abs().mean((2, 3), keepdim=True), atol=1e-6, rtol=1e-5)
    assert torch.allclose(pix_c, pix_n, atol=1e-6, rtol=1e-5) and torch.allclose(ch_c, ch_n, atol=1e-6, rtol=1e-5)
    Sn = S.clone().requires_grad_(True)
    out_n = masked_mse_sums(Sn, T, wf, wb, wp, cc)
    out_c = masked_mse_sums(Scl, Tcl, wf, wb, wp, cc)
    assert torch.equal(out_c, masked_mse_sums(Scl, Tcl, wf, wb, wp, cc))          # fixed summation order
    sq = (S - T).double() ** 2
    ref = torch.stack([(sq * wf.double()).sum(), (sq * wb.double()).sum(),
                       (sq * wp.double() * cc.double().view(B, C, 1, 1)).sum()])
    assert float(((out_c.double() - ref).abs() / ref.abs()).max()) < 2e-6
    assert float(((out_c - out_n).abs() / out_n.abs()).max()) < 2e-6
    wts = torch.tensor([0.3, 1.7, -0.9], device=dev)
    (out_n * wts).sum().backward(); (out_c * wts).sum().backward()
    assert Scl.grad.is_contiguous(memory_format=torch.channels_last)
    assert float((Scl.grad - Sn.grad).abs().max()) <= 1e-6 * float(Sn.grad.abs().max())
    # without the third term / channel factors
    o2 = masked_mse_sums(Scl, Tcl, wf, wb)
    assert float(o2[2]) == 0.0 and abs(float(o2[0]) - float(ref[0])) < 2e-6 * float(ref[0])


def test_channel_mean_fused_into_attention_and_dS_kernels():
    """spatial term of the FGD loss (bevdet_distill.py:1272-1278): mean_c of teacher / student comes out of the |x|-mean
    pass, and the gradient of the student's mean is added inside the masked-MSE dS kernel."""
    from distill_bev_amd.distill_loss import abs_mean_maps, masked_mse_sums
    dev = _dev()
    B, C, H, W = 2, 384, 32, 24
    g = torch.Generator().manual_seed(1)
    S = torch.randn((B, C, H, W), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    T = torch.randn((B, C, H, W), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    wf = torch.rand((B, 1, H, W), generator=g).to(dev); wb = torch.rand((B, 1, H, W), generator=g).to(dev)
    gmap = torch.randn((B, 1, H, W), generator=g).to(dev)
    pix, ch, pool = abs_mean_maps(S, with_pool=True)
    assert torch.allclose(pool, S.mean(1, keepdim=True), atol=1e-6)
    assert torch.allclose(pix, S.abs().mean(1, keepdim=True), atol=1e-6, rtol=1e-5)
    assert abs_mean_maps(S.contiguous(), with_pool=True)[2] is None          # NCHW: caller uses torch.mean
    Sa = S.clone().requires_grad_(True)
    sums, sp = masked_mse_sums(Sa, T, wf, wb, None, None, pool)
    (sums[0] * 0.3 + sums[1] * 1.1 + (sp * gmap).sum()).backward()
    Sb = S.clone().requires_grad_(True)
    sums_b = masked_mse_sums(Sb, T, wf, wb)
    (sums_b[0] * 0.3 + sums_b[1] * 1.1 + (Sb.mean(1, keepdim=True) * gmap).sum()).backward()
    assert torch.equal(sums, sums_b)
    assert float((Sa.grad - Sb.grad).abs().max()) <= 2e-6 * float(Sb.grad.abs().max())
    # pooled output unused downstream: gradient of the sums alone is unchanged
    Sc = S.clone().requires_grad_(True)
    sums_c, _ = masked_mse_sums(Sc, T, wf, wb, None, None, pool)
    sums_c[0].backward()
    Sd = S.clone().requires_grad_(True)
    masked_mse_sums(Sd, T, wf, wb)[0].backward()
    assert torch.equal(Sc.grad, Sd.grad)


def test_masked_mse_determinism_and_small_odd_channels():
    from distill_bev_amd.distill_loss import masked_mse_sums
    dev = _dev()
    B, C, H, W = 3, 37, 8, 12
    S = torch.randn((B, C, H, W), device=dev); T = torch.randn((B, C, H, W), device=dev)
    wf = torch.rand((B, 1, H, W), device=dev); wb = torch.rand((B, 1, H, W), device=dev)
    a = masked_mse_sums(S, T, wf, wb)
    b = masked_mse_sums(S, T, wf, wb)
    assert torch.equal(a, b) and float(a[2]) == 0.0
    sq = (S - T).double() ** 2
    assert abs(float(a[0]) - float((sq * wf.double()).sum())) < 1e-3
    assert abs(float(a[1]) - float((sq * wb.double()).sum())) < 1e-3


@pytest.mark.parametrize("shape,scale", [((2, 256, 32, 32), 4), ((1, 6, 5, 7), 4), ((2, 3, 1, 9), 2), ((1, 8, 16, 16), 3)])
@pytest.mark.parametrize("channels_last", [False, True])
def test_upsample_bilinear_align_corners_vs_torch(shape, scale, channels_last):
    """vs torch's own upsample_bilinear2d (the op the reference calls) on the CPU, forward and backward."""
    from distill_bev_amd.distill_loss import UpsampleBilinearAC
    dev = _dev()
    torch.manual_seed(0)
    x_cpu = torch.randn(shape, requires_grad=True)
    ref = torch.nn.functional.interpolate(x_cpu, scale_factor=scale, mode="bilinear", align_corners=True)
    g = torch.randn_like(ref)
    ref.backward(g)
    x = x_cpu.detach().to(dev)
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    y = UpsampleBilinearAC(scale)(x)
    assert y.shape == ref.shape
    # the source coordinate r*o carries 1 ulp (1.9e-6 at o~127) of rounding freedom -> 1e-5 on O(1) data
    assert float((y.detach().cpu() - ref.detach()).abs().max()) < 2e-5
    gg = g.to(dev)
    if channels_last:
        gg = gg.contiguous(memory_format=torch.channels_last)
    y.backward(gg)
    assert float((x.grad.cpu() - x_cpu.grad).abs().max()) < 1e-4
