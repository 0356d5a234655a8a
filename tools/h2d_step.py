#!/usr/bin/env python3
"""PCIe-inclusive step time: the synthetic batch lives in pinned host memory and is uploaded every step
(same stream, non_blocking) before the training step (dev tool for the DESIGN.md note)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distill_bev_amd.train_step import Trainer, build_model, make_batch

dev = torch.device("cuda:0")
model, cfg = build_model(allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, channels_last=True)
batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
host_imgs = tuple(t.cpu().pin_memory() for t in batch["img_inputs"])
host_pts = [p.cpu().pin_memory() for p in batch["points"]]
nbytes = sum(t.numel() * t.element_size() for t in host_imgs) + sum(p.numel() * 4 for p in host_pts)


def upload():
    b = dict(batch)
    b["img_inputs"] = tuple(t.to(dev, non_blocking=True) for t in host_imgs)
    b["points"] = [p.to(dev, non_blocking=True) for p in host_pts]
    return b


for _ in range(4):
    tr.step(upload())
torch.cuda.synchronize()
for label, fn in (("resident", lambda: tr.step(batch)), ("upload every step", lambda: tr.step(upload()))):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(f"{label:18s}: {(time.perf_counter() - t) / 10 * 1e3:7.1f} ms/step   ({nbytes / 1e6:.0f} MB per step from pinned host memory)")
