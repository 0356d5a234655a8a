"""optim.MultiTensorAdamW (csrc/adamw.hip): one launch per parameter group, the clip factor applied as the gradient is read -- against
torch.optim.AdamW(fused=True) behind torch.nn.utils.clip_grad_norm_ over several steps on tensors of every kind the detector has
(channels-last 4-D filters, vectors, odd lengths), state_dict round trips between the two, and the fall-backs."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(64, 3, 7, 7), (128, 64, 1, 1), (64, 64, 3, 3), (256,), (1,), (33, 17), (5000,), (4097,), (2, 3, 5, 7)]


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    ps = []
    for i, sh in enumerate(SHAPES):
        t = torch.randn(sh, generator=g).to(dev)
        if len(sh) == 4 and i % 2 == 0:
            t = t.contiguous(memory_format=torch.channels_last)
        ps.append(torch.nn.Parameter(t))
    return ps


def _grads(ps, step, scale):
    g = torch.Generator().manual_seed(100 + step)
    for p in ps:
        gr = (torch.randn(p.shape, generator=g) * scale).to(p.device)
        p.grad = gr.contiguous(memory_format=torch.channels_last) if (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last)
                                                                      and not p.is_contiguous()) else gr


@pytest.mark.parametrize("clip", [None, 35.0, 0.5])
def test_steps_match_torch_fused_adamw_behind_clip_grad_norm(clip):
    from distill_bev_amd.optim import MultiTensorAdamW, clip_factor
    dev = torch.device("cuda:0")
    a, b = _params(dev, 1), _params(dev, 1)
    kw = dict(lr=2e-3, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8)
    oa = torch.optim.AdamW(a, fused=True, **kw)
    ob = MultiTensorAdamW(b, **kw)
    worst = 0.0
    for step in range(6):
        _grads(a, step, 3.0); _grads(b, step, 3.0)
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(a, clip)
            tot, c = clip_factor(b, clip)
            assert c.device.type == "cuda" and c.dim() == 0
            ob.step(grad_scale=c)
        else:
            ob.step()
        oa.step()
        for pa, pb in zip(a, b):
            sa, sb = oa.state[pa], ob.state[pb]
            assert float(sa["step"]) == float(sb["step"]) == step + 1
            for x, y in ((pa, pb), (sa["exp_avg"], sb["exp_avg"]), (sa["exp_avg_sq"], sb["exp_avg_sq"])):
                assert x.stride() == y.stride()
                d = float((x.detach() - y.detach()).abs().max() / x.detach().abs().max().clamp_min(1e-30))
                worst = max(worst, d)
    assert ob.multi_launches == 6 and ob.torch_steps == 0
    assert worst <= 2.5e-7, worst                                  # <= ~2 ulp of fp32 on any tensor after six steps (0: bit-identical)
    print("max relative deviation from torch's fused AdamW:", worst)


def test_state_dicts_are_interchangeable_and_fallbacks_step_with_torch():
    from distill_bev_amd.optim import MultiTensorAdamW
    dev = torch.device("cuda:0")
    a, b = _params(dev, 2), _params(dev, 2)
    kw = dict(lr=1e-3, weight_decay=0.05)
    oa = torch.optim.AdamW(a, fused=True, **kw)
    ob = MultiTensorAdamW(b, **kw)
    for step in range(2):
        _grads(a, step, 1.0); _grads(b, step, 1.0)
        oa.step(); ob.step()
    # torch's checkpoint into ours and back
    ob2 = MultiTensorAdamW(_params(dev, 2), **kw)
    for p, q in zip(ob2.param_groups[0]["params"], a):
        p.data.copy_(q.data)
    import copy
    ob2.load_state_dict(copy.deepcopy(oa.state_dict()))       # (a live state_dict aliases the optimizer's tensors; a file does not)
    oa2 = torch.optim.AdamW([torch.nn.Parameter(p.detach().clone()) for p in b], fused=True, **kw)
    oa2.load_state_dict(copy.deepcopy(ob.state_dict()))
    c = ob2.param_groups[0]["params"]
    _grads(a, 7, 1.0); _grads(c, 7, 1.0)
    oa.step(); ob2.step()
    assert ob2.multi_launches == 1
    for pa, pc in zip(a, c):
        assert torch.allclose(pa, pc, rtol=1e-6, atol=1e-7) and float(ob2.state[pc]["step"]) == 3
    # an option the kernel does not implement (amsgrad): the whole step is torch's, the factor applied in place first
    p = torch.nn.Parameter(torch.randn((8, 4, 3, 3), device=dev).contiguous(memory_format=torch.channels_last))
    q = torch.nn.Parameter(p.detach().clone())
    o1, o2 = MultiTensorAdamW([p], amsgrad=True, **kw), torch.optim.AdamW([q], fused=True, amsgrad=True, **kw)
    g = torch.randn((8, 4, 3, 3), device=dev).contiguous(memory_format=torch.channels_last)
    p.grad, q.grad = g.clone(), g.clone() * 0.5
    o1.step(grad_scale=torch.tensor(0.5, device=dev)); o2.step()
    assert o1.torch_steps == 1 and o1.multi_launches == 0 and torch.equal(p, q)
    assert torch.equal(p.grad, g * 0.5)


def test_rejects_bad_arguments():
    from distill_bev_amd import _lib as L
    assert L.lib().dbev_adamw_multi(None, None, 3, None, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0.1, 0.1, None) != 0
    assert L.lib().dbev_adamw_multi(None, None, 0, None, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0.1, 0.1, None) == 0
