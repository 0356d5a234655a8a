#!/usr/bin/env python3
"""Dense (GEMM-shaped) FLOPs of one bench-configuration training step, from forward hooks on every Conv2d /
ConvTranspose2d / Linear (+ the DCN column GEMM): forward FLOPs, x3 where autograd is recording (data + weight
gradient), x1 under no_grad (teacher, adjacent frame)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn as nn
from distill_bev_amd.train_step import Trainer, build_model, make_batch

dev = torch.device("cuda:0")
model, cfg = build_model(allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, channels_last=True)
batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
tot = {"fwd": 0.0, "bwd": 0.0}
by = {}
pending = {}


def hook(m, inp, out):
    x = inp[0]
    if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        kh, kw = m.kernel_size
        if isinstance(m, nn.ConvTranspose2d):
            fl = 2.0 * x.shape[0] * x.shape[2] * x.shape[3] * m.in_channels * m.out_channels * kh * kw / m.groups
        else:
            fl = 2.0 * out.shape[0] * out.shape[2] * out.shape[3] * m.in_channels * m.out_channels * kh * kw / m.groups
    else:
        fl = 2.0 * x.numel() / x.shape[-1] * m.in_features * m.out_features
    tot["fwd"] += fl
    by[type(m).__name__ + " fwd"] = by.get(type(m).__name__ + " fwd", 0.0) + fl
    pending.setdefault(id(m), []).append((fl, bool(x.requires_grad)))


def bhook(m, gin, gout):
    # fires once per forward call that takes part in backward: weight gradient always, data gradient if the input needs one
    fl, need_dx = pending[id(m)].pop()
    add = fl * (2.0 if need_dx else 1.0)
    tot["bwd"] += add
    by[type(m).__name__ + " bwd"] = by.get(type(m).__name__ + " bwd", 0.0) + add


mods = [tr.detector, tr.detector.teacher_model]
for r in mods:
    for m in r.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
            m.register_forward_hook(hook)
            m.register_full_backward_hook(bhook)
tr.step(batch)
torch.cuda.synchronize()
dcn = 2.0 * 48 * 16 * 44 * 2304 * 256            # column GEMM of the DCNv2 layer (F.conv2d inside dcn.py, not a module)
tot["fwd"] += 2 * dcn; tot["bwd"] += 2 * dcn       # two frames forward, one of them backward (dgrad + wgrad)
total = tot["fwd"] + tot["bwd"]
print({k: round(v / 1e12, 3) for k, v in by.items()})
print(f"forward {tot['fwd']/1e12:.3f} TFLOP, backward {tot['bwd']/1e12:.3f} TFLOP; per step {total/1e12:.2f} TFLOP")
for ms in (115.4, 166.0):
    print(f"   / {ms} ms = {total / ms / 1e9:.1f} TFLOP/s")
