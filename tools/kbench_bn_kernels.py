#!/usr/bin/env python3
"""Every kernel of the fused BatchNorm passes IN ISOLATION (library event log) at the step's activation shapes: achieved TB/s per
kernel, to set against the box's streaming ceilings (tools/membench.hip) and against the in-step figures of bench.py (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from distill_bev_amd import _lib as L
from distill_bev_amd import bn_act as BA

dev = torch.device("cuda:0")
for (N, C, H, W, res) in [(48, 64, 64, 176, False), (48, 256, 64, 176, True), (48, 128, 32, 88, False), (48, 512, 32, 88, True),
                          (48, 256, 16, 44, False), (48, 1024, 16, 44, True), (48, 2048, 8, 22, True), (8, 2048, 128, 128, False)]:
    bn = nn.BatchNorm2d(C).to(dev).train()
    x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn_like(x).requires_grad_(True) if res else None
    g = torch.randn_like(x)
    for it in range(6):
        if it == 2:
            L.kernel_timing(True); L.kernel_timing_read()
        y = BA.bn_act(x, bn, r, True)
        y.backward(g)
        x.grad = None; bn.zero_grad(set_to_none=True)
        if r is not None:
            r.grad = None
    log = L.kernel_timing_read()
    L.kernel_timing(False)
    mb = x.numel() * 4 / 1e6
    print(f"N={N} C={C} {H}x{W} res={int(res)}  {mb:.0f} MB: " + "  ".join(
        f"{k} {sum(b for _, b in v) / sum(m for m, _ in v) / 1e9:.2f} TB/s ({1e3 * sum(m for m, _ in v) / len(v):.0f} us)" for k, v in log.items()))
