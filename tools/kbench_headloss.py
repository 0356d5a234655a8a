#!/usr/bin/env python3
"""GPU time of CenterHead.loss (targets + focal + L1 terms, 6 tasks) forward+backward on fake predictions (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile
from distill_bev_amd import synthetic as syn
from distill_bev_amd.center_head import LiDARBoxes
from distill_bev_amd.train_step import build_model

dev = torch.device("cuda:0")
m, _ = build_model(allow_synthetic_teacher=True)
head = m.pts_bbox_head.to(dev)
rng = np.random.default_rng(0)
B = 8
boxes, labels = [], []
for _ in range(B):
    b, lab = syn.gt_boxes(30, rng)
    boxes.append(LiDARBoxes(b)); labels.append(torch.from_numpy(lab))
chans = dict(reg=2, height=1, dim=3, rot=2, vel=2)


def preds():
    out = []
    for names in head.class_names:
        d = {k: torch.randn((B, c, 128, 128), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) * 1.0
             for k, c in chans.items()}
        d["heatmap"] = torch.randn((B, len(names), 128, 128), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) * 1.0
        out.append([d])
    return tuple(out)


def run():
    p = preds()
    losses = head.loss(boxes, labels, p)
    sum(losses.values()).backward()


for _ in range(3):
    run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        run()
    torch.cuda.synchronize()
ka = prof.key_averages()
dev_total = sum(e.self_device_time_total for e in ka)
n_kern = sum(e.count for e in ka if e.self_device_time_total > 0)
randn = sum(e.self_device_time_total for e in ka if "randn" in e.key or "normal_" in e.key)
print(f"CenterHead.loss fwd+bwd: {dev_total/5/1e3:.2f} ms GPU per call (of which fake-pred generation {randn/5/1e3:.2f} ms), {n_kern/5:.0f} device ops")
