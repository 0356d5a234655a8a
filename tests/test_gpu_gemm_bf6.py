"""fp32 GEMM of the 1x1 convolutions on the bf16 matrix cores (csrc/gemm_bf6.hip, distill_bev_amd/gemm_bf6.py) vs the op the reference
runs (nn.Conv2d(k=1) -> the library's fp32 convolution; res_block.py:102-230, necks/fpn.py:10-204).  The claim under test: fp32 accuracy
-- the error against an fp64 GEMM is at or below the error of the library's own fp32 kernel on the same inputs (bound: 1.25 x the
library's error + 1e-7 of the output scale), for the forward product and for both gradients the module hands back."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _mk(n, ci, co, h, w, seed, relu=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((n, ci, h, w), generator=g)
    x = (torch.relu(x) if relu else x).to(DEV).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn((co, ci, 1, 1), generator=g) / ci ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    return x, wt


def _err(y, ref):
    return float((y.double() - ref).abs().max()) / float(ref.abs().max())


SHAPES = [(2, 64, 64, 8, 8),            # one group of four chunks, one 64-column block
          (4, 64, 256, 16, 16),         # 128-column blocks
          (3, 192, 192, 16, 16),        # three groups, N = 3 x 64
          (2, 256, 64, 32, 16),         # four groups of the matrix accumulator
          (1, 2048, 128, 16, 16),       # deep reduction (32 groups)
          (2, 1024, 512, 16, 8)]


@pytest.mark.parametrize("n,ci,co,h,w", SHAPES)
def test_forward_is_as_exact_as_the_fp32_library_kernel(n, ci, co, h, w, monkeypatch):
    from distill_bev_amd import gemm_bf6 as G
    monkeypatch.setattr(G, "_MIN_ITEMS", 1)
    x, wt = _mk(n, ci, co, h, w, 11)
    assert G.eligible(x, wt)
    y = G.product(x, wt)
    lib = F.conv2d(x, wt)
    ref = F.conv2d(x.double(), wt.double())
    assert y.shape == lib.shape and y.is_contiguous(memory_format=torch.channels_last)
    e, el = _err(y, ref), _err(lib, ref)
    assert e <= 1.25 * el + 1e-7, (e, el)
    assert e < 1e-6
    assert torch.equal(y, G.product(x, wt))           # deterministic


def test_signed_inputs_and_extreme_magnitudes(monkeypatch):
    """operands spanning 2^-60 ... 2^60: the split must not lose the small ones (every piece keeps its own exponent)"""
    from distill_bev_amd import gemm_bf6 as G
    monkeypatch.setattr(G, "_MIN_ITEMS", 1)
    g = torch.Generator().manual_seed(3)
    x = (torch.randn((2, 64, 8, 8), generator=g) * torch.exp2(torch.randint(-60, 60, (2, 1, 8, 8), generator=g).float()))   # K = 64
    x = x.to(DEV).contiguous(memory_format=torch.channels_last)
    wt = torch.randn((64, 64, 1, 1), generator=g).to(DEV)
    y = G.product(x, wt)
    ref = F.conv2d(x.double(), wt.double())
    rowscale = ref.abs().amax(dim=1, keepdim=True).clamp(min=1e-300)
    assert float(((y.double() - ref).abs() / rowscale).max()) < 2e-6     # relative to each pixel's own scale
    z = G.product(torch.zeros_like(x), wt)
    assert float(z.abs().max()) == 0.0


@pytest.mark.parametrize("n,ci,co,h,w", [(4, 64, 256, 16, 16), (2, 256, 64, 32, 16), (2, 128, 128, 16, 16), (3, 256, 384, 16, 24),
                                         (2, 1024, 128, 8, 16), (2, 64, 64, 32, 32), (2, 192, 320, 16, 16),
                                         # round 6: pixel counts that are no multiple of 128 (the BEVFormer recipe's 58 x 100 / 29 x 50 maps):
                                         (3, 256, 128, 29, 50), (1, 128, 256, 58, 100), (2, 64, 192, 7, 9), (1, 320, 64, 5, 27)])
def test_module_gradients_vs_fp64(n, ci, co, h, w, monkeypatch):
    """forward, data gradient and weight gradient (all four tile shapes: 128 / 64 output x 128 / 64 input channels) on the bf16x6 kernels"""
    from distill_bev_amd import gemm_bf6 as G
    monkeypatch.setattr(G, "_MIN_ITEMS", 1)
    monkeypatch.setattr(G, "_MIN_WGRAD_ROWS", 1)
    torch.manual_seed(7)
    m = nn.Sequential(nn.Conv2d(ci, co, 1, bias=False)).to(DEV).to(memory_format=torch.channels_last)
    assert G.use_bf6_convs(m) == 1 and type(m[0]) is G.Bf6Conv2d and G.use_bf6_convs(m) == 0
    assert list(m.state_dict()) == ["0.weight"]
    x, _ = _mk(n, ci, co, h, w, 5, relu=False)
    xa = x.clone().requires_grad_(True)
    y = m(xa)
    gy = torch.randn_like(y)
    y.backward(gy)
    xd = x.double().requires_grad_(True)
    wd = m[0].weight.detach().double().requires_grad_(True)
    yd = F.conv2d(xd, wd)
    yd.backward(gy.double())
    # the library's fp32 gradients on the same inputs, for the bound
    xl = x.clone().requires_grad_(True)
    wl = m[0].weight.detach().clone().requires_grad_(True)
    F.conv2d(xl, wl).backward(gy)
    assert _err(y, yd.detach()) <= 1.25 * _err(F.conv2d(x, wl.detach()), yd.detach()) + 1e-7
    assert _err(xa.grad, xd.grad) <= 1.25 * _err(xl.grad, xd.grad) + 1e-7
    assert _err(m[0].weight.grad, wd.grad) <= 1.25 * _err(wl.grad, wd.grad) + 1e-7
    assert m[0].weight.grad.shape == m[0].weight.shape
    if ci % 64 == 0 and co % 64 == 0:                             # the bf16x6 weight gradient ran (128- or 64-wide tiles): fixed summation order
        g1 = G.weight_gradient(x, gy, m[0].weight)
        assert g1 is not None and torch.equal(g1, G.weight_gradient(x, gy, m[0].weight))
        assert torch.equal(g1.reshape(co, ci), m[0].weight.grad.reshape(co, ci))
    with torch.no_grad():
        assert torch.equal(m(x), y.detach())                     # the no-grad path runs the same kernel


def test_small_layers_stay_with_the_library_and_packs_follow_the_weight():
    from distill_bev_amd import gemm_bf6 as G
    x, wt = _mk(1, 64, 64, 16, 16, 1)                            # two tiles: below _MIN_ITEMS
    assert not G.eligible(x, wt)
    m = nn.Conv2d(64, 64, 1, bias=False).to(DEV).to(memory_format=torch.channels_last)
    m.__class__ = G.Bf6Conv2d
    assert torch.equal(m(x), F.conv2d(x, m.weight))
    p1 = G.packed(m.weight)
    assert G.packed(m.weight) is p1
    before = p1.clone()
    with torch.no_grad():
        m.weight.mul_(2.0)                                       # version counter moves: packed again -- into the SAME buffer (round 6)
    assert G.packed(m.weight) is p1 and not torch.equal(p1, before)
    from distill_bev_amd.bn_act import invalidate_eval_coef
    p2 = G.packed(m.weight)
    m.weight.data.mul_(0.5)                                      # a write the version counter does not see ...
    invalidate_eval_coef(m)                                      # ... is followed by the documented invalidation
    assert G.packed(m.weight) is not p2


def test_weight_gradient_full_size_share_merge(monkeypatch):
    """a ResNet stage-3 layer at the step's size: 33 792 pixels in 32 shares (dbev_gemm_bf16x6_backward_weight + the share sum)"""
    from distill_bev_amd import gemm_bf6 as G
    g = torch.Generator().manual_seed(9)
    x = torch.relu(torch.randn((48, 256, 16, 44), generator=g)).to(DEV).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((48, 1024, 16, 44), generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    w = torch.zeros((1024, 256, 1, 1), device=DEV)
    gw = G.weight_gradient(x, gy, w)
    lib = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    ref = gy.permute(0, 2, 3, 1).reshape(-1, 1024).double().t() @ x.permute(0, 2, 3, 1).reshape(-1, 256).double()
    e, el = _err(gw.reshape(1024, 256), ref), _err(lib.reshape(1024, 256), ref)
    assert e <= 1.25 * el + 1e-7 and e < 1e-6, (e, el)


def test_packs_follow_a_fused_optimizer_step(monkeypatch):
    """torch.optim.AdamW(fused=True) -- the Trainer's optimizer -- updates weights without moving `_version`; the kept bf16 planes and
    Winograd packs must still follow the new weights (the post-step hook of _lib.ensure_param_version_hook)"""
    from distill_bev_amd import gemm_bf6 as G
    from distill_bev_amd import wino
    monkeypatch.setattr(G, "_MIN_ITEMS", 1)
    monkeypatch.setattr(wino, "_MIN_WG", 1)
    torch.manual_seed(0)
    m = nn.Sequential(nn.Conv2d(64, 64, 1, bias=False), nn.Conv2d(64, 64, 3, padding=1, bias=False)).to(DEV).to(memory_format=torch.channels_last)
    assert G.use_bf6_convs(m) == 1 and wino.use_wino_convs(m) == 1
    opt = torch.optim.AdamW(m.parameters(), lr=0.05, fused=True)
    x = torch.randn((4, 64, 16, 16), device=DEV).contiguous(memory_format=torch.channels_last)
    for _ in range(2):
        y = m(x)
        ref = F.conv2d(F.conv2d(x.double(), m[0].weight.detach().double()), m[1].weight.detach().double(), padding=1)
        assert float((y.detach().double() - ref).abs().max()) < 1e-5 * float(ref.abs().max()), "forward ran on stale packed weights"
        opt.zero_grad(set_to_none=True)
        y.square().mean().backward()
        opt.step()
    with torch.no_grad():                                        # the no-grad paths keep packs too
        y = m(x)
        ref = F.conv2d(F.conv2d(x.double(), m[0].weight.double()), m[1].weight.double(), padding=1)
        assert float((y.double() - ref).abs().max()) < 1e-5 * float(ref.abs().max())


def test_biased_layers_and_the_dcn_contraction_take_the_same_kernels(monkeypatch):
    """colsum.BiasSumConv2d (1x1 with bias), colsum.conv_bn_cancelled_bias and dcn.modulated_deform_conv2d_raw hand eligible GEMMs to
    the bf16x6 kernels: outputs and gradients against the same calls with the path switched off (the library's fp32 kernels)"""
    from distill_bev_amd import colsum, dcn
    from distill_bev_amd import gemm_bf6 as G
    monkeypatch.setattr(G, "_MIN_ITEMS", 1)
    monkeypatch.setattr(G, "_MIN_WGRAD_ROWS", 1)
    torch.manual_seed(2)
    conv = nn.Conv2d(128, 192, 1).to(DEV).to(memory_format=torch.channels_last)
    conv.__class__ = colsum.BiasSumConv2d
    x = torch.randn((2, 128, 16, 16), device=DEV).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((2, 192, 16, 16), device=DEV).contiguous(memory_format=torch.channels_last)

    def run(on):
        monkeypatch.setattr(G, "_ON", on)
        xa = x.clone().requires_grad_(True)
        conv.zero_grad(set_to_none=True)
        y = conv(xa)
        y.backward(gy)
        return y.detach(), xa.grad, conv.weight.grad.clone(), conv.bias.grad.clone()

    a, b = run(True), run(False)
    ref = F.conv2d(x.double(), conv.weight.detach().double(), conv.bias.detach().double())
    assert _err(a[0], ref) <= 1.25 * _err(b[0], ref) + 1e-7
    for u, v in zip(a[1:], b[1:]):
        assert float((u - v).abs().max()) <= 2e-6 * float(v.abs().max())
    with torch.no_grad():
        monkeypatch.setattr(G, "_ON", True)
        assert torch.equal(conv(x), a[0])
    # DCNv2: 3x3 deformable convolution 64 -> 64 whose column contraction (K = 576) is a 1x1 layer
    w = (torch.randn((64, 64, 3, 3), device=DEV) / 24.0).requires_grad_(True)
    xd = torch.randn((2, 64, 16, 16), device=DEV).contiguous(memory_format=torch.channels_last)
    om = (0.5 * torch.randn((2, 27, 16, 16), device=DEV)).contiguous(memory_format=torch.channels_last)
    outs = []
    for on in (True, False):
        monkeypatch.setattr(G, "_ON", on)
        xr = xd.clone().requires_grad_(True)
        y = dcn.modulated_deform_conv2d_raw(xr, om, w, None)
        gx, gw = torch.autograd.grad(y, (xr, w), torch.ones_like(y))
        outs.append((y.detach(), gx, gw))
    for u, v in zip(*outs):
        assert float((u - v).abs().max()) <= 3e-6 * float(v.abs().max())


def test_few_rows_take_64_column_tiles():
    """the 8 x 22 maps of the last ResNet stage (66 row blocks): 64-column tiles double the workgroups; pack and launch agree on the width"""
    from distill_bev_amd import gemm_bf6 as G
    assert G.tile_n(8448, 512) == 64 and G.tile_n(33792, 1024) == 128 and G.tile_n(8448, 192) == 64
    assert G.shape_ok(8448, 2048, 512) and not G.shape_ok(8448, 2048, 256)
    x, wt = _mk(48, 2048, 512, 8, 22, 4)
    y = G.product(x, wt)
    ref = F.conv2d(x.double(), wt.double())
    e, el = _err(y, ref), _err(F.conv2d(x, wt), ref)
    assert e <= 1.25 * el + 1e-7 and e < 1e-6, (e, el)
    gx = G.data_gradient(y, wt)                                 # [M, 512] x [512, 2048]: 128-column tiles, the transposed pack
    refg = F.conv_transpose2d(y.double(), wt.double())
    assert gx is not None and _err(gx, refg) < 1e-6


@pytest.mark.parametrize("n,ci,co,h,w", [(4, 64, 256, 16, 16), (2, 256, 64, 32, 16), (48, 2048, 512, 8, 22)])
def test_statistics_rows_from_the_epilogue(n, ci, co, h, w, monkeypatch):
    """conv1x1_stats: the partial (sum y, sum y^2) rows per 128-row block equal the column sums of the output it wrote, and a
    training-mode norm fed with them (bn_act(..., pre=rows)) gives what it gives when it takes its own statistics pass"""
    from distill_bev_amd import gemm_bf6 as G
    from distill_bev_amd.bn_act import bn_act
    monkeypatch.setattr(G, "_MIN_ITEMS", 1)
    x, wt = _mk(n, ci, co, h, w, 21)
    y, part = G.conv1x1_stats(x, wt)
    M = n * h * w
    assert part.shape == (M // 128, 2, co) and torch.equal(y, G.product(x, wt))
    y2 = y.permute(0, 2, 3, 1).reshape(M // 128, 128, co).double()
    assert float((part[:, 0].double() - y2.sum(1)).abs().max()) <= 1e-5 * float(y2.abs().sum(1).max())
    assert float((part[:, 1].double() - (y2 * y2).sum(1)).abs().max()) <= 1e-5 * float((y2 * y2).sum(1).max())
    torch.manual_seed(1)
    bn_a, bn_b = nn.BatchNorm2d(co).to(DEV).train(), nn.BatchNorm2d(co).to(DEV).train()
    oa, ob = bn_act(y, bn_a, None, True, pre=part), bn_act(y, bn_b, None, True)
    assert float((oa - ob).abs().max()) <= 2e-5 * max(1.0, float(ob.abs().max()))
    assert float((bn_a.running_var - bn_b.running_var).abs().max()) < 1e-5 and float((bn_a.running_mean - bn_b.running_mean).abs().max()) < 1e-6
    # differentiable: the second output carries no gradient
    xa = x.clone().requires_grad_(True)
    wa = wt.clone().requires_grad_(True)
    ya, pa = G.conv1x1_stats(xa, wa)
    assert not pa.requires_grad
    gx, gw = torch.autograd.grad(ya, (xa, wa), torch.ones_like(ya))
    assert gx.shape == x.shape and gw.shape == wt.shape


def test_huge_finite_values_stay_finite_and_non_finite_inputs_propagate(monkeypatch):
    """a finite fp32 value above the largest bf16 (3.3895e38 < |x| <= 3.4028e38) must not become inf / NaN in the split
    (activations: split on the fly in b6_fwd / b6_wgrad; weights: b6_pack) -- the library's fp32 GEMM returns a finite result there;
    inf / NaN inputs still reach the output as non-finite values (a training loop's non-finite check must keep working)."""
    from distill_bev_amd import gemm_bf6 as G
    monkeypatch.setattr(G, "_MIN_ITEMS", 1)
    monkeypatch.setattr(G, "_MIN_WGRAD_ROWS", 1)
    big = 3.4e38
    x = torch.zeros((2, 64, 8, 8), device=DEV).contiguous(memory_format=torch.channels_last)
    x[0, 3, 1, 1], x[1, 5, 2, 2], x[0, 7, 4, 4] = big, -big, 0.5
    wt = torch.zeros((64, 64, 1, 1), device=DEV)
    wt[2, 3], wt[2, 5], wt[4, 7], wt[6, 7] = 0.25, 0.5, big, -big          # huge activation x small weight, small activation x huge weight
    y = G.product(x, wt)
    ref = F.conv2d(x.double(), wt.double())
    assert bool(torch.isfinite(y).all())
    nz = ref != 0
    assert int(nz.sum()) == 4
    assert float(((y.double() - ref)[nz] / ref[nz]).abs().max()) < 1e-6
    # weight gradient (both operands split on the fly): dW[o, c] = sum_m gy[m, o] x[m, c]
    gy = torch.zeros((2, 64, 8, 8), device=DEV).contiguous(memory_format=torch.channels_last)
    gy[0, 9, 1, 1] = 0.5
    gw = G.weight_gradient(x, gy, wt)
    assert gw is not None and bool(torch.isfinite(gw).all())
    assert abs(float(gw[9, 3, 0, 0]) / (0.5 * big) - 1.0) < 1e-6
    # non-finite inputs propagate
    x2 = x.clone(); x2[0, 3, 1, 1] = float("inf")
    assert not bool(torch.isfinite(G.product(x2, wt)[0, :, 1, 1]).all())
    x3 = x.clone(); x3[0, 3, 1, 1] = float("nan")
    assert bool(torch.isnan(G.product(x3, wt)[0, 2, 1, 1]))


# ---- stride-2 1x1 convolutions: subsample + GEMM (csrc/stride2.hip) -------------------------------------------------------------------
def test_subsample2_and_its_gradient_are_bit_exact():
    """dbev_subsample2_nhwc == x[:, :, ::2, ::2]; dbev_upsample2_zero_nhwc == its autograd gradient (values at the even pixels, zeros
    elsewhere, every element written)"""
    from distill_bev_amd import gemm_bf6 as G
    g = torch.Generator().manual_seed(5)
    for (n, c, h, w) in [(2, 64, 8, 12), (3, 256, 16, 44), (1, 4, 2, 2), (2, 128, 6, 10)]:
        x = torch.randn((n, c, h, w), generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
        y = G.subsample2(x)
        assert y.is_contiguous(memory_format=torch.channels_last) and torch.equal(y, x[:, :, ::2, ::2])
        gy = torch.randn((n, c, h // 2, w // 2), generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
        gx = G.upsample2_zero(gy, h, w)
        ref = torch.zeros_like(x)
        ref[:, :, ::2, ::2] = gy
        assert gx.is_contiguous(memory_format=torch.channels_last) and torch.equal(gx, ref)


@pytest.mark.parametrize("n,ci,co,h,w", [(4, 64, 128, 16, 32), (2, 256, 512, 16, 16), (3, 128, 64, 32, 16)])
def test_stride2_module_vs_fp64(n, ci, co, h, w, monkeypatch):
    """Bf6Conv2d(stride=2): forward, data gradient (zeros at the skipped pixels) and weight gradient vs fp64 F.conv2d(stride=2) -- the
    `downsample` convolution of a stage-first bottleneck (res_block.py:102-230); error bound of the stride-1 layers"""
    from distill_bev_amd import gemm_bf6 as G
    monkeypatch.setattr(G, "_MIN_ITEMS", 1)
    monkeypatch.setattr(G, "_MIN_WGRAD_ROWS", 1)
    x, wt = _mk(n, ci, co, h, w, 23, relu=False)
    conv = nn.Conv2d(ci, co, 1, stride=2, bias=False).to(DEV)
    conv.weight.data.copy_(wt)
    assert G.use_bf6_convs(conv) == 1 and type(conv) is G.Bf6Conv2d
    assert G.eligible_s2(x, conv.weight)
    xg = x.clone().requires_grad_(True)
    y = conv(xg)
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(2)).to(DEV).contiguous(memory_format=torch.channels_last)
    y.backward(gy)
    xd = x.double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True)
    yr = F.conv2d(xd, wd, stride=2)
    yr.backward(gy.double())
    lib = F.conv2d(x, conv.weight, stride=2)
    e, el = _err(y.detach(), yr.detach()), _err(lib.detach(), yr.detach())
    assert e <= 1.25 * el + 1e-7 and e < 1e-6, (e, el)
    assert _err(xg.grad, xd.grad) < 1e-6 and _err(conv.weight.grad, wd.grad) < 2e-6
    assert float(xg.grad[:, :, 1::2, :].abs().max()) == 0.0 and float(xg.grad[:, :, :, 1::2].abs().max()) == 0.0
    with torch.no_grad():
        assert torch.equal(conv(x), y.detach())               # the no-grad route (detached frame): same kernels
    # statistics epilogue of the strided layer = statistics of its output
    z, part = G.conv1x1_s2_stats(x, conv.weight)
    assert torch.equal(z, y.detach())
    s1 = part[:, 0].double().sum(0)
    assert torch.allclose(s1, z.double().sum((0, 2, 3)), rtol=1e-6, atol=1e-4)
    # odd sizes stay with the library
    xo = x[:, :, : h - 1].contiguous(memory_format=torch.channels_last)
    assert not G.eligible_s2(xo, conv.weight)
    assert torch.allclose(conv(xo), F.conv2d(xo, conv.weight, stride=2), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n,ci,co,h,w", [(18, 256, 1024, 29, 50), (3, 64, 128, 7, 9), (1, 128, 64, 1, 129), (2, 512, 256, 58, 100)])
def test_rows_past_the_last_block_load_zeros_and_store_nothing(n, ci, co, h, w, monkeypatch):
    """round 6: M = N H W of any size.  The last 128-row block is read and written through buffer descriptors of its valid rows: the output
    (and the memory BEHIND it: a guard band) is what the library computes / untouched, the statistics rows count ceil(M / 128) and sum
    to the column sums of y exactly as for full blocks, and the weight gradient reads pixels past M as zeros"""
    from distill_bev_amd import gemm_bf6 as G, _lib as L
    monkeypatch.setattr(G, "_MIN_ITEMS", 1)
    monkeypatch.setattr(G, "_MIN_WGRAD_ROWS", 1)
    x, wt = _mk(n, ci, co, h, w, 13)
    M = n * h * w
    assert M % 128 != 0 and G.eligible(x, wt)
    y, part = G.product(x, wt, stats=True)
    ref = F.conv2d(x.double(), wt.double())
    lib = F.conv2d(x, wt)
    assert _err(y, ref) <= 1.25 * _err(lib, ref) + 1e-7
    assert part.shape[0] == -(-M // 128)
    s = part.double().sum(0)
    yy = y.double().permute(0, 2, 3, 1).reshape(-1, co)
    assert torch.allclose(s[0], yy.sum(0), rtol=1e-5, atol=1e-3) and torch.allclose(s[1], (yy * yy).sum(0), rtol=1e-5, atol=1e-3)
    # guard band: the launch writes M rows and not one float more
    tn = G.tile_n(M, co)
    buf = torch.full((M * co + 128 * co,), 7.25, device=DEV)
    with torch.cuda.device(DEV):
        L.call("dbev_gemm_bf16x6_forward_stats", L.ptr(x), L.ptr(G.packed(wt, False, tn)), L.ptr(buf), None, M, ci, co, ci, int(tn), L.stream_ptr(DEV))
    assert torch.equal(buf[:M * co].view(n, h, w, co).permute(0, 3, 1, 2), y) and bool((buf[M * co:] == 7.25).all())
    # weight gradient: fixed order, equal to fp64 within the library's error
    gy = torch.randn_like(y)
    g1 = G.weight_gradient(x, gy, wt)
    assert g1 is not None and torch.equal(g1, G.weight_gradient(x, gy, wt))
    gref = torch.einsum("nohw,nihw->oi", gy.double(), x.double())
    glib = torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    assert _err(g1.reshape(co, ci), gref) <= 1.25 * _err(glib.reshape(co, ci), gref) + 1e-7


@pytest.mark.parametrize("shape,ci,co", [((40000,), 256, 512), ((2, 20000), 512, 256), ((1, 40000), 256, 128), ((3, 20001), 256, 64)])
def test_linear_on_tokens_vs_fp64(shape, ci, co, monkeypatch):
    """round 6: nn.Linear on [tokens, C] (the BEVFormer encoder's projections and FFNs, transformer_modules/*.py) re-classed to
    Bf6Linear: output, input gradient, weight gradient and bias gradient against fp64, no worse than torch's fp32 linear; few tokens
    (the decoder's 900 queries) stay with the library; same parameters and state-dict keys"""
    from distill_bev_amd import gemm_bf6 as G
    torch.manual_seed(5)
    lin = nn.Sequential(nn.Linear(ci, co)).to(DEV)
    ref = nn.Sequential(nn.Linear(ci, co)).to(DEV)
    ref.load_state_dict(lin.state_dict())
    assert G.use_bf6_linears(lin) == 1 and type(lin[0]) is G.Bf6Linear and G.use_bf6_linears(lin) == 0
    assert list(lin.state_dict()) == ["0.weight", "0.bias"]
    x = torch.randn(shape + (ci,), device=DEV)
    assert G.eligible_linear(x, lin[0].weight)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y, yr = lin(xa), ref(xb)
    gy = torch.randn_like(y)
    y.backward(gy); yr.backward(gy)
    xd = x.double().requires_grad_(True)
    wd, bd = ref[0].weight.detach().double().requires_grad_(True), ref[0].bias.detach().double().requires_grad_(True)
    yd = F.linear(xd, wd, bd)
    yd.backward(gy.double())
    assert y.shape == yr.shape
    for got, lib, want in ((y.detach(), yr.detach(), yd.detach()), (xa.grad, xb.grad, xd.grad), (lin[0].weight.grad, ref[0].weight.grad, wd.grad),
                           (lin[0].bias.grad, ref[0].bias.grad, bd.grad)):
        assert _err(got, want) <= 1.25 * _err(lib, want) + 2e-7, (_err(got, want), _err(lib, want))
    with torch.no_grad():
        assert torch.equal(lin(x), y.detach())
    z = torch.relu_(lin(x.clone().requires_grad_(True)))                       # an in-place op on the output (the FFN's ReLU(inplace=True))
    z.sum().backward()
    few = torch.randn((900, ci), device=DEV)                                   # the decoder's query count: below the fill threshold
    assert not G.eligible_linear(few, lin[0].weight)
    assert torch.equal(lin(few), ref(few))
