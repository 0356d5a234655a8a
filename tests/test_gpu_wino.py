"""Winograd F(2x2, 3x3) convolution kernels (csrc/wino.hip) vs torch's convolution in fp64 (the op the reference's dense blocks run:
nn.Conv2d(k=3, s=1, p=1), res_block.py:11-230): forward, data gradient, statistics rows, bias, every tile-block shape, partial
blocks at the image edges.  Tolerance: 2e-5 of the output scale (fp32 Winograd: 2-4x a direct fp32 convolution's error, measured
against the same fp64 reference and asserted below)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _mk(N, C, Co, H, W, seed, bias):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, C, H, W), generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Co, C, 3, 3), generator=g) / (3.0 * C ** 0.5)).to(DEV).contiguous(memory_format=torch.channels_last)
    b = torch.randn((Co,), generator=g).to(DEV) if bias else None
    return x, w, b


SHAPES = [(2, 16, 64, 16, 16, False),      # one 8x8 tile block per image, a single patch stage
          (3, 64, 64, 16, 44, True),        # 8 x 22 tiles: partial blocks along W (8,8)
          (2, 32, 128, 8, 64, False),       # 4 x 32 tiles: the (4,16) block shape
          (1, 48, 64, 6, 10, True),         # partial blocks both ways, three k stages
          (2, 256, 256, 16, 12, False),     # deep reduction (32 k groups)
          (1, 64, 192, 34, 18, True),       # odd tile counts, three channel blocks
          (1, 4, 64, 2, 2, False),          # a single tile, one k group of 4 (no 8-channel format: wino_fwd3 only)
          (5, 12, 64, 4, 6, True),          # Cin % 8 != 0
          (1, 20, 128, 10, 2, False)]       # one tile column


@pytest.mark.parametrize("N,C,Co,H,W,bias", SHAPES)
def test_forward_and_statistics_vs_fp64(N, C, Co, H, W, bias):
    """(the library picks wino_fwd3 for these small grids; tests/test_gpu_wino.py::test_both_forward_kernels covers wino_fwd)"""
    from distill_bev_amd import wino
    x, w, b = _mk(N, C, Co, H, W, 1, bias)
    assert wino.eligible(x, w)
    y, part = wino.conv3x3_stats(x, w, b)
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), 1, 1)
    scale = float(ref.abs().max())
    err = float((y.double() - ref).abs().max()) / scale
    direct = float((F.conv2d(x, w, b, 1, 1).double() - ref).abs().max()) / scale
    assert err <= 2e-5, err
    assert err <= max(8.0 * direct, 2e-6), (err, direct)          # same error class as the direct fp32 kernel
    assert y.is_contiguous(memory_format=torch.channels_last)
    # statistics rows: per channel sum y, sum y^2 over all pixels (fixed order: bit-identical when repeated)
    s = part.double().sum(0)
    assert torch.allclose(s[0], y.double().sum((0, 2, 3)), rtol=1e-5, atol=1e-4 * scale)
    assert torch.allclose(s[1], (y.double() ** 2).sum((0, 2, 3)), rtol=1e-5, atol=1e-4 * scale)
    y2, part2 = wino.conv3x3_stats(x, w, b)
    assert torch.equal(y, y2) and torch.equal(part, part2)
    assert torch.equal(wino.conv3x3(x, w, b), y)


@pytest.mark.parametrize("N,C,Co,H,W,bias", SHAPES)
def test_gradients_vs_fp64(N, C, Co, H, W, bias):
    from distill_bev_amd import wino
    x, w, b = _mk(N, C, Co, H, W, 2, bias)
    x.requires_grad_(True); w.requires_grad_(True)
    if b is not None:
        b.requires_grad_(True)
    gy = torch.randn((N, Co, H, W), generator=torch.Generator().manual_seed(3)).to(DEV).contiguous(memory_format=torch.channels_last)
    ins = (x, w) + ((b,) if b is not None else ())
    got = torch.autograd.grad(wino.conv3x3(x, w, b), ins, gy)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    bd = None if b is None else b.detach().double().requires_grad_(True)
    ref = torch.autograd.grad(F.conv2d(xd, wd, bd, 1, 1), (xd, wd) + ((bd,) if bd is not None else ()), gy.double())
    for a, r, name in zip(got, ref, ("grad_x", "grad_w", "grad_b")):
        e = float((a.double() - r).abs().max()) / float(r.abs().max())
        assert e <= 3e-5, (name, e)
    assert got[0].is_contiguous(memory_format=torch.channels_last)


def test_filter_pack_is_the_winograd_transform_of_each_filter():
    """packed[(jb, kg, p, ni, lane, e)] = (G g G^T)[p] of filter (k, j) in the two kernels' consumption orders; both modes
    (forward / rotated-transposed)"""
    from distill_bev_amd import wino
    Co, C = 128, 24
    g = torch.Generator().manual_seed(5)
    w = torch.randn((Co, C, 3, 3), generator=g).to(DEV)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    for mode, wm in ((False, w), (True, w.flip(2, 3).transpose(0, 1))):
        if mode and C % 64:
            continue
        K, J = wm.shape[1], wm.shape[0]
        U = torch.einsum("ia,jkab,cb->jkic", G, wm.double().cpu(), G).reshape(J, K, 16)        # [j, k, p]
        both = wino.pack_filters(w, mode).cpu().double()
        assert both.numel() == 16 * K * J * (2 if K % 8 == 0 else 1)
        packed = both[:16 * K * J].reshape(J // 64, K // 4, 16, 2, 64, 2)            # the format of wino_fwd3 (k groups of 4)
        jb, kg, p, ni, lane, e = np.ix_(*[np.arange(n) for n in packed.shape])
        k = 4 * kg + 2 * (lane >> 5) + e
        j = 64 * jb + 32 * ni + (lane & 31)
        assert np.abs(packed.numpy() - U.numpy()[j, k, p]).max() <= 1e-6
        if K % 8 == 0:                                                               # then the format of wino_fwd (k groups of 8)
            packed = both[16 * K * J:].reshape(J // 64, K // 8, 16, 2, 64, 4)
            jb, kg, p, ni, lane, e = np.ix_(*[np.arange(n) for n in packed.shape])
            k = 8 * kg + 4 * (lane >> 5) + e
            j = 64 * jb + 32 * ni + (lane & 31)
            assert np.abs(packed.numpy() - U.numpy()[j, k, p]).max() <= 1e-6


def test_paired_pack_matches_the_single_direction_packs_and_is_reused():
    """dbev_wino_filter_pack_pair writes, in one launch, exactly what the two single-direction packs write in the slot of the kernel
    each direction gets; the pair stays attached to the weight, and is written again in place when the weight's version changes"""
    from distill_bev_amd import wino, _lib as L
    for (N, C, Co, H, W) in [(2, 64, 128, 16, 16), (48, 64, 64, 16, 44), (1, 64, 64, 8, 8)]:
        x, w, _ = _mk(N, C, Co, H, W, 21, False)
        fwd, dg = wino.packed_pair(w, x.shape, True)
        for pack, mode, K, J in ((fwd, False, C, Co), (dg, True, Co, C)):
            ver = int(L.call("dbev_wino_conv3x3_forward_kernel", N, H, W, K, J))
            ref = wino.pack_filters(w, mode)
            one = 16 * K * J
            sl = slice(one, 2 * one) if ver == 2 else slice(0, one)
            assert torch.equal(pack[sl], ref[sl])
        again = wino.packed_pair(w, x.shape, True)
        assert again[0] is fwd and again[1] is dg                       # same version: the attached pair
        before = fwd[sl].clone()
        w.add_(1.0)                                                     # in-place update (the optimizer's): new version ...
        f2, d2 = wino.packed_pair(w, x.shape, True)
        # ... re-packed INTO THE SAME BUFFERS (round 6: a captured hipGraph may be reading through their addresses, graphed.py)
        assert f2 is fwd and d2 is dg and not torch.equal(f2[sl], before)
        y = wino.conv3x3(x, w)
        assert torch.allclose(y, F.conv2d(x, w, None, 1, 1), atol=2e-5 * float(y.abs().max()))
    # forward first without the data gradient, then the missing direction alone
    x, w, _ = _mk(2, 64, 64, 16, 16, 22, False)
    f, d = wino.packed_pair(w, x.shape, False)
    assert d is None
    f3, d3 = wino.packed_pair(w, x.shape, True)
    assert f3 is f and d3 is not None


def test_frozen_conv_norm_relu_stack_is_one_launch_per_pair():
    """conv -> eval-mode BatchNorm -> ReLU of a frozen nn.Sequential (the teacher's SECOND stages, second.py:60-78) with the norm folded into
    the Winograd launch (wino.ConvNormSequential) == the three stock modules; in training mode / under autograd the stack runs unfolded"""
    import torch.nn as nn
    from distill_bev_amd import wino, bn_act, _lib as L
    torch.manual_seed(3)
    seq = nn.Sequential(nn.Conv2d(64, 64, 3, 2, 1, bias=False), nn.BatchNorm2d(64, eps=1e-3), nn.ReLU(inplace=True),
                        nn.Conv2d(64, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64, eps=1e-3), nn.ReLU(inplace=True),
                        nn.Conv2d(64, 128, 3, 1, 1, bias=True), nn.BatchNorm2d(128, eps=1e-3), nn.ReLU(inplace=True)).to(DEV)
    for m in seq.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.normal_(); m.running_var.uniform_(0.5, 2.0); m.weight.data.normal_(1.0, 0.3); m.bias.data.normal_()
    seq = seq.to(memory_format=torch.channels_last).eval()
    import copy
    ref = copy.deepcopy(seq)
    assert bn_act.fuse_bn_relu_modules(seq) == 3 and wino.use_wino_convs(seq) == 2 and wino.link_conv_norm_stacks(seq) == 2
    assert type(seq) is wino.ConvNormSequential and list(seq.state_dict()) == list(ref.state_dict())
    x = torch.randn((8, 64, 128, 128), device=DEV).contiguous(memory_format=torch.channels_last)     # 64 x 64 behind the strided layer
    with torch.no_grad():
        y, yr = seq(x), ref(x)
    assert y.shape == yr.shape and float((y - yr).abs().max()) <= 2e-5 * float(yr.abs().max())
    assert "_dbev_wino_folded" in seq[3].__dict__ and "_dbev_wino_folded" in seq[6].__dict__          # both pairs took the folded path
    # a changed norm -> new coefficients -> new fold
    with torch.no_grad():
        seq[4].weight.mul_(0.5); ref[4].weight.mul_(0.5)
        y, yr = seq(x), ref(x)
    assert float((y - yr).abs().max()) <= 2e-5 * float(yr.abs().max())
    # writes through .data move no version counter: the documented remedy forgets the kept packs / folds
    seq[3].weight.data.mul_(2.0); ref[3].weight.data.mul_(2.0)
    bn_act.invalidate_eval_coef(seq)
    with torch.no_grad():
        y, yr = seq(x), ref(x)
    assert float((y - yr).abs().max()) <= 2e-5 * float(yr.abs().max())
    # training mode: batch statistics, no folding; gradients flow
    seq.train(); ref.train()
    xg = x.clone().requires_grad_(True)
    out = seq(xg)
    out.sum().backward()
    outr = ref(x)
    assert float((out.detach() - outr.detach()).abs().max()) <= 1e-4 * float(outr.detach().abs().max()) and xg.grad is not None


_LAYER_CODE = """
import sys, torch, torch.nn.functional as F
from distill_bev_amd import wino, _lib as L
dev = torch.device('cuda:0')
N, C, Co, H, W = 48, 64, 256, 16, 44
g = torch.Generator().manual_seed(31)
x = torch.randn((N, C, H, W), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn((Co, C, 3, 3), generator=g) / (3.0 * C ** 0.5)).to(dev).contiguous(memory_format=torch.channels_last)
b = torch.randn((Co,), generator=g).to(dev)
code = int(L.call("dbev_wino_conv3x3_forward_kernel", N, H, W, C, Co))
y, part = wino.conv_packed(x, wino.pack_filters(w, False, x.shape), Co, b, stats=True)
ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
scale = float(ref.abs().max())
assert float((y.double() - ref).abs().max()) <= 2e-5 * scale
assert part.shape[0] == wino.stats_rows(x.shape, Co)
s = part.double().sum(0)
assert torch.allclose(s[0], ref.sum((0, 2, 3)), rtol=1e-5, atol=1e-3 * scale)
assert torch.allclose(s[1], (ref ** 2).sum((0, 2, 3)), rtol=1e-5, atol=1e-3 * scale * scale)
for _ in range(3):                                     # work items are handed out dynamically: the result must not depend on who got which
    y2, part2 = wino.conv_packed(x, wino.pack_filters(w, False, x.shape), Co, b, stats=True)
    assert torch.equal(y, y2) and torch.equal(part, part2)
xg = x.clone().requires_grad_(True)
wg = w.clone().requires_grad_(True)
wino.conv3x3(xg, wg, b).backward(torch.ones_like(y))
xr = x.double().requires_grad_(True)
wr = w.double().requires_grad_(True)
F.conv2d(xr, wr, b.double(), 1, 1).backward(torch.ones_like(ref))
assert float((xg.grad.double() - xr.grad).abs().max()) <= 2e-5 * float(xr.grad.abs().max())
assert float((wg.grad.double() - wr.grad).abs().max()) <= 2e-5 * float(wr.grad.abs().max())
torch.save(dict(y=y.cpu(), part_sum=part.double().sum(0).cpu(), gx=xg.grad.cpu()), sys.argv[1])
print('OK code', code)
"""


def test_persistent_hybrid_and_plain_launches_of_a_2_25_round_layer(tmp_path):
    """576 work items = 2.25 rounds of 256 CUs, three ways (the choice is read once per process: a subprocess each): persistent
    workgroups walking per-XCD item queues (wino_fwdp, kernel code 2: DBEV_WINO_PERSIST=1; measured neutral, off by default), the
    default hybrid (wino_fwd for the whole rounds, wino_fwd3 for the remaining tile rows, code 6) and plain wino_fwd
    (DBEV_WINO_FWD_V=2): each against fp64
    (output, statistics rows, both gradients), repeated launches bit-identical, the persistent kernel's output bit-identical to plain
    wino_fwd's (the per-item arithmetic is the same whichever workgroup runs it), the hybrid's equal to rounding"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("DBEV_WINO_FWD_V", "DBEV_WINO_PERSIST", "DBEV_WINO_HYBRID")}
    outs = {}
    for name, env, code in (("persistent", {"DBEV_WINO_PERSIST": "1"}, 2), ("hybrid", {}, 6), ("plain", {"DBEV_WINO_FWD_V": "2"}, 2)):
        f = str(tmp_path / (name + ".pt"))
        r = subprocess.run([sys.executable, "-c", _LAYER_CODE, f], env=dict(base, **env), capture_output=True, text=True, cwd=root)
        assert r.returncode == 0 and "OK code %d" % code in r.stdout, (name, r.stdout[-500:], r.stderr[-1500:])
        outs[name] = torch.load(f)
    assert torch.equal(outs["persistent"]["y"], outs["plain"]["y"])      # (the data gradient's 144 items go to another kernel by default)
    # (the hybrid's tail rows run on wino_fwd3, whose two position halves meet in a different summation order: equal to rounding)
    d = (outs["hybrid"]["y"] - outs["plain"]["y"]).abs().max()
    assert float(d) <= 2e-6 * float(outs["plain"]["y"].abs().max()) and not torch.equal(outs["hybrid"]["y"], torch.zeros_like(outs["hybrid"]["y"]))
    assert torch.allclose(outs["persistent"]["part_sum"], outs["hybrid"]["part_sum"], rtol=1e-6)      # (row layouts differ: the sums agree)


@pytest.mark.parametrize("block", ["basic", "bottleneck"])
def test_residual_blocks_on_the_winograd_kernels_match_the_stock_convolutions(block, monkeypatch):
    """use_wino_convs re-classes the 3x3 convolutions of a residual block (res_block.py:11-230); the block then runs the Winograd
    forward with the norm's statistics from its epilogue, the Winograd data gradient and weight gradient.  Against the same block on
    the library's convolutions (same weights, training mode): output, running statistics, input and parameter gradients."""
    import copy
    from distill_bev_amd import nets, wino
    monkeypatch.setattr(wino, "_MIN_WG", 0)                       # small test maps: take the kernels whatever the grid size
    torch.manual_seed(11)
    if block == "basic":
        ref = nets.BasicBlock(64, 64)
        x = torch.randn(2, 64, 16, 24)
    else:
        ds = torch.nn.Sequential(torch.nn.Conv2d(128, 256, 1, bias=False), torch.nn.BatchNorm2d(256))
        ref = nets.Bottleneck(128, 64, downsample=ds)
        x = torch.randn(2, 128, 12, 16)
    ref = ref.to(DEV).train()
    for m in ref.modules():
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    new = copy.deepcopy(ref)
    assert wino.use_wino_convs(new) == (2 if block == "basic" else 1)
    assert wino.use_wino_convs(new) == 0                          # idempotent
    assert list(new.state_dict()) == list(ref.state_dict())
    x = x.to(DEV).contiguous(memory_format=torch.channels_last)
    g = torch.randn_like(x if block == "basic" else torch.empty(2, 256, 12, 16, device=DEV)).contiguous(memory_format=torch.channels_last)
    outs = []
    for m, flag in ((ref, "0"), (new, "1")):
        monkeypatch.setenv("DBEV_WINO", flag)
        xi = x.clone().requires_grad_(True)
        y = m(xi)
        grads = torch.autograd.grad(y, [xi] + list(m.parameters()), g)
        outs.append((y.detach(), grads, [b.clone() for b in m.buffers()]))
    (y0, g0, b0), (y1, g1, b1) = outs
    scale = float(y0.abs().max())
    assert float((y0 - y1).abs().max()) <= 2e-5 * scale
    for a, b in zip(g0, g1):
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()), (a.shape, float((a - b).abs().max()) / float(a.abs().max()))
    for a, b in zip(b0, b1):
        assert torch.allclose(a.float(), b.float(), rtol=1e-5, atol=1e-6)


def test_small_grids_stay_on_the_library_kernel():
    """a layer with too few workgroups for one-workgroup-per-CU kernels is left to the library (wino.worthwhile)"""
    from distill_bev_amd import wino
    conv = wino.WinoConv2d(64, 64, 3, padding=1, bias=False).to(DEV)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    x = torch.randn(1, 64, 16, 16, device=DEV).contiguous(memory_format=torch.channels_last)
    assert wino.eligible(x, conv.weight) and not wino.worthwhile(x, 64)
    assert torch.allclose(conv(x), F.conv2d(x, conv.weight, None, 1, 1), atol=1e-5)


def test_both_forward_kernels():
    """DBEV_WINO_FWD_V forces one of the three forward kernels (2: 64-tile workgroups, one per CU; 3: 32-tile workgroups, two per CU;
    4: persistent 64-tile workgroups walking the item queues -- here with far fewer items than CUs, so most workgroups leave at once
    and queues get emptied by their neighbours); each
    against fp64 on every test shape, in a subprocess each (the choice is read once per process)."""
    import subprocess, sys, os
    code = """
import torch, torch.nn.functional as F
from distill_bev_amd import wino
dev = torch.device('cuda:0')
for N, C, Co, H, W in [(2, 16, 64, 16, 16), (3, 64, 64, 16, 44), (2, 32, 128, 8, 64), (1, 48, 64, 6, 10), (2, 256, 256, 16, 12), (1, 64, 192, 34, 18)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn((N, C, H, W), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Co, C, 3, 3), generator=g) / (3.0 * C ** 0.5)).to(dev).contiguous(memory_format=torch.channels_last)
    y, part = wino.conv3x3_stats(x, w, None)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    assert err <= 2e-5, (N, C, Co, H, W, err)
    assert torch.allclose(part.double().sum(0)[0], y.double().sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
print('OK')
"""
    for v in ("2", "3", "4"):
        env = dict(os.environ, DBEV_WINO_FWD_V=v)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and "OK" in r.stdout, (v, r.stdout[-500:], r.stderr[-1500:])
