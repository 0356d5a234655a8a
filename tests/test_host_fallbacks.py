"""CPU: the host-side mirrors added in round 3 leave CPU tensors to the stock modules (no library call, no ledger note) and keep the
reference's values: stem max pooling, the depth-head tail, forked residual outputs, the pinned-staging upload helper."""
import torch
import torch.nn as nn


def test_max_pool_and_depth_head_take_the_module_sequence_on_cpu():
    from distill_bev_amd.depth_head import depth_head, eligible
    from distill_bev_amd.pool import max_pool
    torch.manual_seed(0)
    pool = nn.MaxPool2d(3, 2, 1)
    x = torch.randn(2, 8, 9, 11)
    assert torch.equal(max_pool(pool, x), pool(x))
    bn, conv = nn.BatchNorm2d(8).train(), nn.Conv2d(8, 5, 1)
    state = {k: v.clone() for k, v in bn.state_dict().items()}
    assert not eligible(x, bn, conv)
    digit, prob = depth_head(x, bn, conv)
    bn.load_state_dict(state)
    ref = conv(bn(x))
    assert torch.allclose(digit, ref, atol=1e-6) and torch.allclose(prob, ref.softmax(dim=1), atol=1e-6)
    assert torch.allclose(prob.sum(dim=1), torch.ones_like(prob.sum(dim=1)), atol=1e-6)


def test_forked_is_the_identity_without_the_fused_path_and_h2d_is_a_plain_copy_on_cpu():
    from distill_bev_amd import _lib as L
    from distill_bev_amd import bn_act as BA
    x = torch.randn(2, 8, 5, 5, requires_grad=True)
    bn = nn.BatchNorm2d(8).train()
    y = BA.bn_act(x, bn, None, True, fork=True)                 # CPU tensor: stock ops, no second handle
    assert BA.forked(y) is y and not hasattr(y, "_dbev_fork")
    ref = torch.relu(nn.BatchNorm2d(8).train()(x))
    assert torch.allclose(y, ref, atol=1e-6)
    t = torch.arange(6, dtype=torch.int32)
    assert torch.equal(L.h2d(t, "cpu"), t)
    like = L.h2d_like(torch.zeros(3, dtype=torch.float64), [1, 2, 3])
    assert like.dtype == torch.float64 and like.tolist() == [1.0, 2.0, 3.0]


def test_bias_sum_conv_is_the_stock_module_on_the_host():
    """colsum.BiasSumConv2d / the cancelled-bias path only engage for channels-last device tensors (no CPU compute in the product)"""
    import torch
    import torch.nn as nn
    from distill_bev_amd.colsum import BiasSumConv2d, cancelled_bias_ready, use_bias_sum_convs
    torch.manual_seed(0)
    m = nn.Sequential(nn.Conv2d(4, 6, 3, 1, 1), nn.Conv2d(6, 2, 1, bias=False))
    ref = nn.Conv2d(4, 6, 3, 1, 1)
    ref.load_state_dict(m[0].state_dict())
    assert use_bias_sum_convs(m) == 1 and type(m[0]) is BiasSumConv2d and type(m[1]) is nn.Conv2d
    x = torch.randn(2, 4, 5, 5, requires_grad=True)
    assert torch.equal(m[0](x), ref(x))
    assert not cancelled_bias_ready(m[0], nn.BatchNorm2d(6), x)


def test_fused_optimizer_steps_move_the_version_counter():
    """torch's fused AdamW leaves `_version` alone; every kept pack of this package follows it (the global post-step hook of _lib)"""
    import torch
    from distill_bev_amd import _lib as L
    from distill_bev_amd import wino  # noqa: F401  (importing the package's cache holders installs the hook)
    L.ensure_param_version_hook()
    p = torch.nn.Parameter(torch.randn(8, 8))
    q = torch.nn.Parameter(torch.randn(4))                      # no gradient: not updated, not bumped
    opt = torch.optim.AdamW([p, q], lr=1e-3, fused=True)
    p.grad = torch.randn_like(p)
    v0, w0, before = p._version, q._version, p.detach().clone()
    opt.step()
    assert not torch.equal(before, p.detach())
    assert p._version > v0 and q._version == w0
    assert len(L._VERSION_HOOK) == 1
