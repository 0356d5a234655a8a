"""CenterHead with its 36 branch stacks evaluated as a few wide convolution / norm calls (distill_bev_amd/head_batch.py) against
the per-branch module calls of the same head (the reference's structure, centerpoint_head.py:17-130,352-363): every branch
output, the gradient of the shared feature map, every parameter gradient, running statistics and num_batches_tracked."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

TASKS = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "construction_vehicle"]),
         dict(num_class=2, class_names=["bus", "trailer"]), dict(num_class=1, class_names=["barrier"]),
         dict(num_class=2, class_names=["motorcycle", "bicycle"]), dict(num_class=2, class_names=["pedestrian", "traffic_cone"])]


def _head(in_channels=96, hidden=64):
    from distill_bev_amd.center_head import CenterHead
    return CenterHead(in_channels=in_channels, tasks=TASKS, common_heads=dict(reg=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
                      share_conv_channel=hidden, separate_head=dict(type="SeparateHead", init_bias=-2.19, final_kernel=3, head_conv=hidden),
                      loss_cls=dict(type="GaussianFocalLoss", reduction="mean"), loss_bbox=dict(type="L1Loss", reduction="mean", loss_weight=0.25))


def _prep(head):
    from distill_bev_amd.bn_act import fuse_bn_relu_modules
    from distill_bev_amd.skinny_conv import use_skinny_convs
    head = head.to(DEV).to(memory_format=torch.channels_last).train()
    fuse_bn_relu_modules(head)
    assert use_skinny_convs(head) == 36
    return head


@pytest.mark.parametrize("hw", [(20, 12), (33, 17)])
def test_batched_branches_match_the_per_branch_calls(hw):
    from distill_bev_amd import head_batch
    torch.manual_seed(3)
    ref = _prep(_head())
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
            m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0)
    bat = copy.deepcopy(ref)
    assert head_batch.plan_branches(bat) == 36
    assert [len(g) for g in bat._branch_plan["groups"]] == [32, 4]      # 2048 + 256 channels
    assert getattr(ref, "_branch_plan", None) is None
    x0 = torch.randn((3, 96, *hw), device=DEV).contiguous(memory_format=torch.channels_last)
    res = []
    for head in (ref, bat):
        x = x0.clone().requires_grad_(True)
        out = head([x])
        probe = torch.empty((3, 64, *hw), device=DEV).contiguous(memory_format=torch.channels_last)
        assert head_batch.applies(head, probe) == (head is bat)
        flat = [(t, k, v) for t, task in enumerate(out) for k, v in task[0].items()]
        g = torch.Generator(device=DEV).manual_seed(11)
        loss = sum((v * torch.randn(v.shape, device=DEV, generator=g)).sum() for _, _, v in flat)
        loss.backward()
        res.append((flat, x.grad, {n: p.grad for n, p in head.named_parameters()}, {n: b.clone() for n, b in head.named_buffers()}))
    (f0, gx0, gp0, b0), (f1, gx1, gp1, b1) = res
    assert [(t, k) for t, k, _ in f0] == [(t, k) for t, k, _ in f1]              # same dict order per task
    rel = lambda a, b: float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)
    for (t, k, a), (_, _, b) in zip(f1, f0):
        assert a.shape == b.shape and a.is_contiguous(memory_format=torch.channels_last)
        assert rel(a, b) <= 2e-5, (t, k, rel(a, b))
    # (the wide convolution and the 36 narrow ones round differently; a ReLU gate of the 3.9 M hidden activations that sits within
    #  that rounding of zero flips, which moves single gradient elements -- hence relative-to-max bounds well above 1e-5)
    assert rel(gx1, gx0) <= 3e-3, rel(gx1, gx0)
    assert set(gp0) == set(gp1)
    for n in gp0:
        assert gp1[n] is not None and gp1[n].shape == gp0[n].shape, n
        assert rel(gp1[n], gp0[n]) <= 3e-3, (n, rel(gp1[n], gp0[n]))
    for n in b0:
        if n.endswith("num_batches_tracked"):
            assert int(b1[n]) == int(b0[n]) == 1, n
        else:
            assert rel(b1[n], b0[n]) <= 1e-5, (n, rel(b1[n], b0[n]))
    # eval / no_grad / pruned calls keep the per-branch modules
    bat.eval()
    with torch.no_grad():
        assert not head_batch.applies(bat, probe)
        o = bat([x0], only=("heatmap",))
        assert list(o[0][0]) == ["heatmap"]


def test_plan_is_refused_for_heads_the_kernels_do_not_cover():
    from distill_bev_amd import head_batch
    h = _head(hidden=16)                       # Ch / 4 = 4 lanes per pixel: below the skinny kernels' range
    h = h.to(DEV).to(memory_format=torch.channels_last)
    from distill_bev_amd.bn_act import fuse_bn_relu_modules
    fuse_bn_relu_modules(h)
    assert head_batch.plan_branches(h) == 0 and h._branch_plan is None
    x = torch.randn((2, 96, 8, 8), device=DEV).contiguous(memory_format=torch.channels_last)
    assert len(h([x])) == 6
