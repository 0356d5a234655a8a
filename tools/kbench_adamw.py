#!/usr/bin/env python3
"""optimizer step of the student's parameter set: torch fused AdamW (+ in-place clip) vs optim.MultiTensorAdamW (clip factor in the step)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distill_bev_amd.optim import MultiTensorAdamW, clip_factor
from distill_bev_amd.train_step import build_model, to_channels_last

dev = torch.device("cuda:0")
model, cfg = build_model(allow_synthetic_teacher=True)
model = model.to(dev)
to_channels_last(model)
params = [p for p in model.parameters() if p.requires_grad]
print(len(params), "tensors,", sum(p.numel() for p in params) / 1e6, "M parameters")
for p in params:
    p.grad = torch.randn_like(p) * 0.01


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


ot = torch.optim.AdamW(params, lr=2e-4, weight_decay=0.01, fused=True)
om = MultiTensorAdamW(params, lr=2e-4, weight_decay=0.01)


def torch_path():
    torch.nn.utils.clip_grad_norm_(params, 35.0)
    ot.step()


def multi_path():
    _t, c = clip_factor(params, 35.0)
    om.step(grad_scale=c)


print("torch: clip_grad_norm_ + fused AdamW  %.3f ms   (step alone %.3f)" % (timed(torch_path), timed(ot.step)))
print("multi: clip_factor + one launch       %.3f ms   (step alone %.3f)" % (timed(multi_path), timed(om.step)))
