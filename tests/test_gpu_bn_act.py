"""Fused BatchNorm2d(+residual)(+ReLU) kernels (csrc/bn_act.hip) through the C ABI vs the torch op sequence
F.batch_norm -> add -> relu evaluated in fp64 on the host.  fp32 kernels: tolerances are a few ulp of the
tensor scale (stated per check)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(x, res, bn, relu, training):
    x = x.detach().double().requires_grad_(True)
    r = None if res is None else res.detach().double().requires_grad_(True)
    w = bn.weight.detach().double().requires_grad_(True)
    b = bn.bias.detach().double().requires_grad_(True)
    rm, rv = bn.running_mean.detach().double().clone(), bn.running_var.detach().double().clone()
    out = F.batch_norm(x, rm, rv, w, b, training, bn.momentum, bn.eps)
    if r is not None:
        out = out + r
    if relu:
        out = F.relu(out)
    return out, (x, r, w, b), (rm, rv)


@pytest.fixture(params=["finalize", "ticket"])
def merge_mode(request, monkeypatch):
    """How the per-workgroup partial rows are merged: by the small finalize launches (default), or by last-arriver tickets in the
    producer + a merge in the consumer's prologue (DBEV_BN_TICKET=1; opt-in, measured slower -- DESIGN.md section 7)."""
    monkeypatch.setenv("DBEV_BN_TICKET", "1" if request.param == "ticket" else "0")
    return request.param


def _close(a, r, tol, what):
    a = a.detach().double().cpu(); r = r.detach().double().cpu()
    scale = float(r.abs().max()) + 1e-12
    err = float((a - r).abs().max()) / scale
    assert err < tol, f"{what}: max err {err:.3e} x scale {scale:.3e}"


@pytest.mark.parametrize("N,C,H,W,res,relu", [
    (3, 64, 9, 7, False, True),      # C/4 = 16 columns, 16 row phases per workgroup
    (2, 256, 16, 11, True, True),    # bottleneck tail: bn -> +identity -> relu
    (2, 8, 5, 5, True, False),       # tiny C, no activation
    (1, 2048, 4, 3, False, True),    # C/4 = 512 -> two column chunks (gridDim.y = 2)
    (4, 128, 33, 17, False, False),  # plain BN
    (2, 1024, 6, 5, True, True),     # C/4 = 256
    (5, 4, 3, 2, False, True),       # C/4 = 1
])
def test_train_forward_backward_and_running_stats(N, C, H, W, res, relu, merge_mode):
    from distill_bev_amd.bn_act import bn_act, eligible
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn((N, C, H, W), generator=g) * 2.0 + 0.7)
    r = torch.randn((N, C, H, W), generator=g) if res else None
    bn = nn.BatchNorm2d(C, eps=1e-3, momentum=0.1)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g))
        bn.running_mean.copy_(torch.randn(C, generator=g)); bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    ref, (rx, rr, rw, rb), (rm, rv) = _ref(x, r, bn, relu, True)
    gout = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(gout)

    bn = bn.to(DEV).train()
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rd = None if r is None else r.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert eligible(xd, bn, rd)
    out = bn_act(xd, bn, rd, relu)
    assert out.is_contiguous(memory_format=torch.channels_last)
    out.backward(gout.float().to(DEV).contiguous(memory_format=torch.channels_last))
    _close(out, ref, 5e-6, "y")
    _close(bn.running_mean, rm, 1e-6, "running_mean")
    _close(bn.running_var, rv, 2e-6, "running_var")
    assert int(bn.num_batches_tracked) == 1
    M = N * H * W
    _close(xd.grad, rx.grad, 2e-5, "grad_x")
    _close(bn.weight.grad, rw.grad, 1e-5 * max(1.0, M ** 0.5 / 8), "grad_gamma")
    _close(bn.bias.grad, rb.grad, 1e-5 * max(1.0, M ** 0.5 / 8), "grad_beta")
    if res:
        _close(rd.grad, rr.grad, 1e-6, "grad_residual")


def test_train_is_bit_reproducible_and_large_shape():
    from distill_bev_amd.bn_act import bn_act
    torch.manual_seed(0)
    bn = nn.BatchNorm2d(256).to(DEV).train()
    x = torch.randn((12, 256, 64, 176), device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    res = torch.randn_like(x)
    outs = []
    for _ in range(2):
        x.grad = None; bn.zero_grad()
        y = bn_act(x, bn, res, True)
        y.backward(torch.ones_like(y))
        outs.append((y.detach().clone(), x.grad.clone(), bn.weight.grad.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    # against torch on the same device (fp32 both sides)
    bn2 = nn.BatchNorm2d(256).to(DEV).train()
    ref = F.relu(bn2(x.detach()) + res)
    assert float((outs[0][0] - ref).abs().max()) < 1e-5


def test_eval_mode_uses_running_stats_and_fallback_paths():
    from distill_bev_amd import bn_act as BA
    torch.manual_seed(1)
    bn = nn.BatchNorm2d(64, eps=1e-3).to(DEV)
    with torch.no_grad():
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_()
    bn.eval()
    x = torch.randn((2, 64, 8, 8), device=DEV).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        assert BA.eligible(x, bn)
        y = BA.bn_act(x, bn, None, True)
        ref = F.relu(bn(x))
    assert float((y - ref).abs().max()) < 2e-6 * float(ref.abs().max())
    # eval + autograd, NCHW input, odd channel count, disabled(): all take the torch ops and agree with them
    xg = x.clone().requires_grad_(True)
    assert not BA.eligible(xg, bn)
    assert not BA.eligible(x.contiguous(), bn)
    assert not BA.eligible(torch.zeros((1, 6, 2, 2), device=DEV).contiguous(memory_format=torch.channels_last),
                           nn.BatchNorm2d(6).to(DEV))
    with BA.disabled():
        assert not BA.eligible(x, bn)
        assert torch.equal(BA.bn_act(x, bn, None, True), ref)


def test_module_surgery_keeps_state_dict_and_matches_unfused_model():
    from distill_bev_amd import bn_act as BA
    from distill_bev_amd.nets import ResNet
    torch.manual_seed(2)
    net = ResNet(depth=50, out_indices=(2, 3), norm_eval=False).to(DEV).to(memory_format=torch.channels_last).train()
    keys = list(net.state_dict().keys())
    seq = nn.Sequential(nn.Conv2d(3, 16, 3, padding=1), nn.BatchNorm2d(16), nn.ReLU(inplace=True)).to(DEV)
    assert BA.fuse_bn_relu_modules(seq) == 1 and isinstance(seq[1], BA.BatchNormAct2d) and isinstance(seq[2], nn.Identity)
    assert BA.fuse_bn_relu_modules(seq) == 0
    assert list(seq.state_dict().keys()) == ["0.weight", "0.bias", "1.weight", "1.bias", "1.running_mean",
                                             "1.running_var", "1.num_batches_tracked"]
    x = torch.randn((2, 3, 64, 96), device=DEV).contiguous(memory_format=torch.channels_last)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    outs_f = net(x)
    loss = sum(o.square().mean() for o in outs_f); loss.backward()
    gf = net.conv1.weight.grad.clone(); rm_f = net.bn1.running_mean.clone() if hasattr(net, "bn1") else None
    net.load_state_dict(sd); net.zero_grad()
    with BA.disabled():
        outs_u = net(x)
        loss_u = sum(o.square().mean() for o in outs_u); loss_u.backward()
    assert list(net.state_dict().keys()) == keys
    for a, b in zip(outs_f, outs_u):
        assert float((a - b).abs().max()) < 2e-4 * float(b.abs().max())
    assert abs(float(loss) - float(loss_u)) < 1e-4 * abs(float(loss_u))
    gu = net.conv1.weight.grad
    assert float((gf - gu).norm() / gu.norm()) < 5e-3      # 53 BN layers of fp32 round-off, random init


def test_edge_cases_single_value_per_channel_and_tiny_tensors():
    from distill_bev_amd import bn_act as BA
    bn = nn.BatchNorm2d(8).to(DEV).train()
    x1 = torch.randn((1, 8, 1, 1), device=DEV).contiguous(memory_format=torch.channels_last)
    assert not BA.eligible(x1, bn)
    with pytest.raises(ValueError):                       # the reference module's own error, not a kernel result
        BA.bn_act(x1, bn, None, True)
    # two values per channel: smallest legal training batch; unbiased running_var = 2 * biased
    x2 = torch.tensor([[1.0], [3.0]], device=DEV).view(2, 1, 1, 1).repeat(1, 8, 1, 1).contiguous(memory_format=torch.channels_last)
    assert BA.eligible(x2, bn)
    y = BA.bn_act(x2, bn, None, False)
    ref = nn.BatchNorm2d(8).to(DEV).train()
    assert torch.allclose(y, ref(x2), atol=1e-6)
    assert torch.allclose(bn.running_var, ref.running_var, atol=1e-7) and torch.allclose(bn.running_mean, ref.running_mean)
    # rows not a multiple of the row phases / unroll, C/4 = 1..512 already covered above; odd M here
    bn3 = nn.BatchNorm2d(64).to(DEV).train(); ref3 = nn.BatchNorm2d(64).to(DEV).train()
    x3 = torch.randn((1, 64, 7, 13), device=DEV).contiguous(memory_format=torch.channels_last)
    assert torch.allclose(BA.bn_act(x3, bn3, None, True), torch.relu(ref3(x3)), atol=2e-6)


@pytest.mark.parametrize("N,C,H,W,relu", [(2, 256, 16, 11, True), (3, 64, 9, 7, True), (1, 2048, 4, 3, True), (2, 512, 7, 5, False)])
def test_dual_norm_add_relu_of_a_stage_first_residual_block(N, C, H, W, relu, merge_mode):
    """bn_act_dual = relu(bn(x) + bn_d(xd)) (dbev_bn_dual_*): output, the two input gradients, the four parameter gradients
    and both sets of running statistics against the fp64 torch sequence; bit-identical when repeated; eval / no-grad calls
    take the two-step path and give the same values."""
    from distill_bev_amd.bn_act import bn_act_dual
    g = torch.Generator().manual_seed(C + W)
    x = torch.randn((N, C, H, W), generator=g) * 1.5 + 0.3
    xd = torch.randn((N, C, H, W), generator=g) * 0.7 - 0.2
    bns = [nn.BatchNorm2d(C, eps=1e-5, momentum=0.1), nn.BatchNorm2d(C, eps=1e-3, momentum=0.05)]
    with torch.no_grad():
        for bn in bns:
            bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g))
            bn.running_mean.copy_(torch.randn(C, generator=g)); bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    r1, (rx, _, rw, rb), (rm1, rv1) = _ref(x, None, bns[0], False, True)
    r2, (rxd, _, rwd, rbd), (rm2, rv2) = _ref(xd, None, bns[1], False, True)
    ref = F.relu(r1 + r2) if relu else r1 + r2
    gout = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(gout)
    bns = [bn.to(DEV).train() for bn in bns]
    init = [{k: v.clone() for k, v in bn.state_dict().items()} for bn in bns]
    a = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = xd.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)

    def run():
        for bn, st in zip(bns, init):
            bn.load_state_dict(st)
        y = bn_act_dual(a, bns[0], b, bns[1], relu)
        grads = torch.autograd.grad(y, [a, b, bns[0].weight, bns[0].bias, bns[1].weight, bns[1].bias], gout.float().to(DEV))
        return y, grads
    y, grads = run()
    assert type(y.grad_fn).__name__ == "_BNDualTrainBackward"
    _close(y, ref, 2e-6, "dual forward")
    for got, want, tol, what in zip(grads, (rx.grad, rxd.grad, rw.grad, rb.grad, rwd.grad, rbd.grad),
                                    (2e-5, 2e-5, 1e-5, 1e-5, 1e-5, 1e-5), ("dx", "dxd", "dgamma", "dbeta", "dgamma_d", "dbeta_d")):
        _close(got, want, tol, what)
    _close(bns[0].running_mean, rm1, 1e-6, "running_mean"); _close(bns[0].running_var, rv1, 1e-6, "running_var")
    _close(bns[1].running_mean, rm2, 1e-6, "running_mean_d"); _close(bns[1].running_var, rv2, 1e-6, "running_var_d")
    assert int(bns[0].num_batches_tracked) == 1 and int(bns[1].num_batches_tracked) == 1
    y2, grads2 = run()
    assert torch.equal(y, y2) and all(torch.equal(p, q) for p, q in zip(grads, grads2))
    with torch.no_grad():                         # statistics path of the two-step fallback (no autograd): same numbers
        for bn, st in zip(bns, init):
            bn.load_state_dict(st)
        y3 = bn_act_dual(a, bns[0], b, bns[1], relu)
    _close(y3, ref, 2e-6, "two-step forward")


@pytest.mark.parametrize("dual", [False, True])
def test_forked_block_output_hands_two_gradient_addends_to_the_fused_backward(dual):
    """A residual block's output feeds the next block's first convolution and its identity branch.  bn_act(..., fork=True) returns the
    output with a second handle (forked(y)); the two gradients then reach the fused norm backward as two addends
    (dbev_bn_act_backward2 / dbev_bn_dual_backward2) instead of being summed by autograd's accumulation pass.  a + b is the same
    float either way: every gradient is BIT-identical to the unforked graph, and the forked graph has no accumulation add."""
    from distill_bev_amd import bn_act as BA
    torch.manual_seed(3)
    N, C, H, W = 2, 64, 9, 12
    mk = lambda: nn.BatchNorm2d(C).to(DEV).train()
    bn0, bnd, bn1, bn2 = mk(), mk(), mk(), mk()
    conv = nn.Conv2d(C, C, 1, bias=False).to(DEV).to(memory_format=torch.channels_last)
    cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last)
    z, zd, r = [cl(torch.randn(N, C, H, W)).requires_grad_(True) for _ in range(3)]
    gout = cl(torch.randn(N, C, H, W))
    params = [z, zd, r, bn0.weight, bn0.bias, bnd.weight, bnd.bias, bn1.weight, bn1.bias, bn2.weight, bn2.bias, conv.weight]
    init = [{k: v.clone() for k, v in m.state_dict().items()} for m in (bn0, bnd, bn1, bn2)]

    def run(fork):
        for m, st in zip((bn0, bnd, bn1, bn2), init):
            m.load_state_dict(st)
        BA._state["fork"] = fork
        try:
            if dual:
                y = BA.bn_act_dual(z, bn0, zd, bnd, True, fork=True)                  # block k: relu(bn(z) + bn_d(zd))
            else:
                y = BA.bn_act(z, bn0, r, True, fork=True)                             # block k: relu(bn(z) + r)
            assert (getattr(y, "_dbev_fork", None) is not None) == fork
            if fork:
                assert BA.forked(y).data_ptr() == y.data_ptr() and BA.forked(y) is not y
            h = BA.bn_act(conv(y), bn1, None, True)                                   # block k+1: first convolution on y ...
            out = BA.bn_act(h, bn2, BA.forked(y), True)                               # ... identity branch on its second handle
            grads = torch.autograd.grad(out, params, gout, allow_unused=True)
            names = set()
            stack, seen = [out.grad_fn], set()
            while stack:
                f = stack.pop()
                if f is None or f in seen:
                    continue
                seen.add(f); names.add(type(f).__name__)
                stack.extend(g for g, _ in f.next_functions)
            return grads, out.detach().clone()
        finally:
            BA._state["fork"] = True
    (ga, oa), (gb, ob) = run(True), run(False)
    assert torch.equal(oa, ob)
    for a, b, p in zip(ga, gb, params):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b), float((a - b).abs().max())
    used = [g is not None for g in ga]
    assert used[0] and used[-1] and (used[1] if dual else used[2])


@pytest.mark.parametrize("which", ["first", "second"])
def test_in_place_write_to_a_forked_output_is_an_autograd_error_not_a_wrong_gradient(which):
    """The two handles of a forked output share storage but not a version counter.  Both are saved by the fused norm's node, so an
    in-place op on EITHER handle (a neck doing `feat += ...` / `relu_` on a stage output) raises autograd's "modified by an inplace
    operation" error in backward instead of silently changing what the other handle's consumer saved."""
    from distill_bev_amd import bn_act as BA
    torch.manual_seed(4)
    N, C, H, W = 2, 64, 5, 6
    bn = nn.BatchNorm2d(C).to(DEV).train()
    cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last)
    z = cl(torch.randn(N, C, H, W)).requires_grad_(True)
    y = BA.bn_act(z, bn, None, True, fork=True)
    y2 = BA.forked(y)
    assert y2 is not y
    out = (y * 2.0).sum() + (y2 * 3.0).sum()
    with torch.no_grad():
        (y if which == "first" else y2).add_(1.0)
    with pytest.raises(RuntimeError, match="inplace|in-place"):
        out.backward()


def test_eval_coefficients_are_kept_on_the_module_and_follow_every_change_of_its_tensors():
    """eval-mode bn_act keeps scale | shift on the module (dbev_bn_infer_coef once, dbev_bn_act_apply per call): the values must follow
    a training step that moves the running statistics, an optimizer-style in-place update of gamma / beta, and load_state_dict."""
    from distill_bev_amd import bn_act as BA
    torch.manual_seed(1)
    bn = nn.BatchNorm2d(32).to(DEV)
    x = torch.randn(2, 32, 6, 7, device=DEV).contiguous(memory_format=torch.channels_last)

    def check():
        bn.eval()
        with torch.no_grad():
            got = BA.bn_act(x, bn, None, True)
            ref = torch.relu(torch.nn.functional.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps))
        assert torch.allclose(got, ref, atol=2e-6), float((got - ref).abs().max())
        return bn.__dict__["_dbev_eval_coef"][1]
    c0 = check()
    assert check() is c0                                        # unchanged module: the kept tensor is reused
    # every change below re-derives the coefficients INTO THE SAME TENSOR (round 6: a captured hipGraph may read through its address,
    # graphed.py) -- check() has compared the values with torch's batch_norm each time
    v0 = c0.clone()
    bn.train(); BA.bn_act(x * 3 + 1, bn, None, True)            # running statistics move (in-kernel update)
    c1 = check()
    assert c1 is c0 and not torch.equal(c1, v0)
    v1 = c1.clone()
    with torch.no_grad():
        bn.weight.mul_(1.5); bn.bias.add_(0.25)                 # optimizer-style in-place parameter update
    c2 = check()
    assert c2 is c0 and not torch.equal(c2, v1)
    v2 = c2.clone()
    bn.load_state_dict({k: torch.rand_like(v.float()).to(v.dtype) + 0.5 if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    assert check() is c0 and not torch.equal(c0, v2)
    BA.invalidate_eval_coef(bn)                                 # the documented invalidation drops the tensor: a new one next time
    assert check() is not c0


@pytest.mark.parametrize("dual", [False, True])
def test_relu_gate_byte_mask_gives_bit_identical_gradients(dual):
    """Round 5: the forward of a residual norm writes its ReLU gate as one byte per four channels and both backward passes read the
    byte instead of the saved output (dbev_bn_act_train_forward_mask / _backward3, the dual entries likewise): every gradient is
    bit-identical to the path that reads the output (DBEV_BN_GATE_MASK=0), also with the gradient arriving as two addends."""
    import torch.nn as nn
    from distill_bev_amd import bn_act as BA
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31)
    shape = (6, 256, 16, 44)
    cl = lambda: torch.randn(shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    x0, r0, gy, gy2 = cl(), cl(), cl(), cl()
    outs = []
    for use_mask in (True, False):
        BA._state["gate_mask"] = use_mask
        try:
            torch.manual_seed(3)
            bn, bnd = nn.BatchNorm2d(shape[1]).to(dev).train(), nn.BatchNorm2d(shape[1]).to(dev).train()
            with torch.no_grad():
                for m in (bn, bnd):
                    m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.3)
            x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
            y = BA.bn_act_dual(x, bn, r, bnd, True, fork=True) if dual else BA.bn_act(x, bn, r, True, fork=True)
            y2 = BA.forked(y)
            assert y2 is not y
            params = [bn.weight, bn.bias] + ([bnd.weight, bnd.bias] if dual else [])
            grads = torch.autograd.grad([y, y2], [x, r] + params, [gy, gy2])
            outs.append([y.detach()] + [t.detach() for t in grads])
        finally:
            BA._state["gate_mask"] = True
    assert len(outs[0]) == len(outs[1]) and all(torch.equal(a, b) for a, b in zip(*outs))
    # and against the unfused op sequence
    x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
    torch.manual_seed(3)
    bn, bnd = nn.BatchNorm2d(shape[1]).to(dev).train(), nn.BatchNorm2d(shape[1]).to(dev).train()
    with torch.no_grad():
        for m in (bn, bnd):
            m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.3)
    ref = torch.relu(bn(x) + (bnd(r) if dual else r))
    gx, gr = torch.autograd.grad(ref, [x, r], gy + gy2)
    assert torch.allclose(outs[0][0], ref.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(outs[0][1], gx, rtol=1e-4, atol=1e-5) and torch.allclose(outs[0][2], gr, rtol=1e-4, atol=1e-5)


_DZ_SCRIPT = r"""
import sys, torch, torch.nn as nn
from distill_bev_amd import bn_act as BA
dev = torch.device("cuda:0")
out = []
for shape, two in (((6, 256, 16, 44), True), ((6, 256, 16, 44), False), ((3, 64, 7, 5), True)):
    g = torch.Generator().manual_seed(47)
    cl = lambda: torch.randn(shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    x, r, gy, gy2 = cl().requires_grad_(True), cl().requires_grad_(True), cl(), cl()
    torch.manual_seed(5)
    bn = nn.BatchNorm2d(shape[1]).to(dev).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
    y = BA.bn_act(x, bn, r, True, fork=two)
    outs, gys = ([y, BA.forked(y)], [gy, gy2]) if two else ([y], [gy])
    out += [t.detach().cpu() for t in torch.autograd.grad(outs, [x, r, bn.weight, bn.bias], gys)]
torch.save(out, sys.argv[1])
"""


def test_residual_backward_that_writes_dz_in_the_reduce_pass_is_bit_identical_to_the_round4_order(tmp_path):
    """Round 5: for a residual norm with ReLU the statistics pass of the backward writes dz = gate * (dy + dy2) into `grad_residual`
    and the dx pass reads it back (7 tensor passes for the pair instead of 8 when two addends arrive); DBEV_BN_DZ_FIRST=0 keeps the
    round-4 order (dx pass recomputes dz).  The switch is read once per process, so each order runs in its own interpreter: all four
    gradients of three cases (two addends, one addend, ragged tiny shape) are bit-identical."""
    import os, subprocess, sys
    res = []
    for v in ("1", "0"):
        p = tmp_path / f"dz{v}.pt"
        env = dict(os.environ, DBEV_BN_DZ_FIRST=v, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        subprocess.run([sys.executable, "-c", _DZ_SCRIPT, str(p)], check=True, env=env, timeout=600)
        res.append(torch.load(p))
    assert len(res[0]) == 12 and all(torch.equal(a, b) for a, b in zip(*res))
    assert all(bool(t.isfinite().all()) and float(t.abs().max()) > 0 for t in res[0])
