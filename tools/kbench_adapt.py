#!/usr/bin/env python3
"""fused 1x1 adaptation + masked-MSE kernel vs the unfused sequence at the head position (dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from distill_bev_amd import _lib as L
from distill_bev_amd.distill_loss import _FusedAdaptMSE, abs_mean_maps, masked_mse_sums
dev = torch.device("cuda:0")
B = 8
x = torch.randn((B, 256, 128, 128), device=dev).contiguous(memory_format=torch.channels_last)
T = torch.randn((B, 384, 128, 128), device=dev).contiguous(memory_format=torch.channels_last)
conv = nn.Conv2d(256, 384, 1).to(dev).to(memory_format=torch.channels_last)
cc = torch.rand((B, 384), device=dev); w = torch.rand((B, 1, 128, 128), device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
with torch.no_grad():
    tf = t(lambda: _FusedAdaptMSE.apply(x, conv.weight, conv.bias, T, cc))
    def unf():
        s = conv(x); abs_mean_maps(s, with_pool=True); masked_mse_sums(s, T, w, w, w, cc)
    tu = t(unf); tc = t(lambda: conv(x))
fl = 2.0 * B * 128 * 128 * 256 * 384
print(f"fused {tf:.1f} us = {fl/tf/1e6:.1f} TFLOP/s ({fl/tf/1e6/157.3:.2f} of fp32 MFMA peak) | unfused conv+abs_mean+mse {tu:.1f} us (conv alone {tc:.1f} us = {fl/tc/1e6:.1f} TFLOP/s)")
