#!/usr/bin/env python3
"""CenterHead forward + backward at the step's shape (8 x 256 x 128 x 128 -> 6 tasks x 6 branches): per-branch module calls vs
the batched branch groups of head_batch.py (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.backends.cudnn.benchmark = True
from distill_bev_amd import head_batch
from distill_bev_amd.bn_act import fuse_bn_relu_modules
from distill_bev_amd.center_head import CenterHead
from distill_bev_amd.skinny_conv import use_skinny_convs

DEV = "cuda:0"
TASKS = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "construction_vehicle"]),
         dict(num_class=2, class_names=["bus", "trailer"]), dict(num_class=1, class_names=["barrier"]),
         dict(num_class=2, class_names=["motorcycle", "bicycle"]), dict(num_class=2, class_names=["pedestrian", "traffic_cone"])]
head = CenterHead(in_channels=256, tasks=TASKS, common_heads=dict(reg=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
                  share_conv_channel=64, separate_head=dict(type="SeparateHead", init_bias=-2.19, final_kernel=3, head_conv=64))
head = head.to(DEV).to(memory_format=torch.channels_last).train()
fuse_bn_relu_modules(head); use_skinny_convs(head)
x = torch.randn((8, 256, 128, 128), device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)


def step():
    out = head([x])
    loss = sum(v.sum() for task in out for v in task[0].values())
    loss.backward()
    x.grad = None
    for p in head.parameters():
        p.grad = None


for mode in ("per-branch", "batched", "per-branch", "batched"):
    head._branch_plan = None
    if mode == "batched":
        assert head_batch.plan_branches(head) == 36
    for _ in range(3):
        step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print(f"{mode:10s} {(time.perf_counter() - t) * 100:.3f} ms per forward + backward")
