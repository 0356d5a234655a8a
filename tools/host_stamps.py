#!/usr/bin/env python3
"""host-side time stamps of back-to-back steps (dev tool): does the host run ahead of the GPU or does something block it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
from distill_bev_amd import distill_loss as DL, detectors as DT, center_head as CH
from distill_bev_amd.train_step import Trainer, build_model, make_batch, parse_losses

dev = torch.device("cuda:0")
model, cfg = build_model(allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, channels_last=True)
batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
log = []
def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); log.append((tag, t, time.perf_counter())); return r
    setattr(obj, name, g)
fgcls = [v for v in vars(DL).values() if isinstance(v, type) and hasattr(v, "_cell_coords")][0]
wrap(fgcls, "__call__", "fg_mask")
wrap(DT.BEVDepth4DDistill if hasattr(DT, "BEVDepth4DDistill") else type(model), "shift_feature", "shift_feature")
wrap(CH.CenterHead, "get_targets_device", "targets")
for _ in range(4):
    tr.step(batch)
torch.cuda.synchronize()
log.clear()
T0 = time.perf_counter()
for i in range(6):
    t = time.perf_counter()
    losses = tr.module(**batch); loss = parse_losses(losses); t1 = time.perf_counter()
    tr.optimizer.zero_grad(set_to_none=True); loss.backward(); t2 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(tr.params, **tr.grad_clip); tr.optimizer.step(); t3 = time.perf_counter()
    log.append((f"step{i} fwd", t, t1)); log.append((f"step{i} bwd", t1, t2)); log.append((f"step{i} opt", t2, t3))
torch.cuda.synchronize()
print(f"total {(time.perf_counter() - T0) / 6 * 1e3:.1f} ms/step")
for tag, a, b in sorted(log, key=lambda r: r[1]):
    print(f"{(a - T0) * 1e3:9.2f} ms  +{(b - a) * 1e3:7.2f} ms  {tag}")
