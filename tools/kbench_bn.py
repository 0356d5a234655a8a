#!/usr/bin/env python3
"""Fused BN(+add)+ReLU vs the torch/MIOpen op sequence at ResNet-50 shapes of the training step (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import torch.nn.functional as F
from distill_bev_amd import bn_act as BA

dev = torch.device("cuda:0")


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for (N, C, H, W, res) in [(96, 64, 64, 176, False), (96, 256, 64, 176, True), (96, 128, 32, 88, False),
                          (96, 512, 32, 88, True), (96, 1024, 16, 44, True), (96, 2048, 8, 22, True),
                          (96, 64, 128, 352, False), (8, 128, 128, 128, False)]:
    bn = nn.BatchNorm2d(C).to(dev).train()
    x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn_like(x).requires_grad_(True) if res else None
    g = torch.randn_like(x)
    mb = x.numel() * 4 / 1e6

    def fused_f():
        return BA.bn_act(x, bn, r, True)

    def torch_f():
        o = bn(x)
        if r is not None:
            o = o + r
        return F.relu(o, inplace=True)

    def fb(f):
        y = f()
        y.backward(g)
        x.grad = None; bn.zero_grad(set_to_none=True)
        if r is not None:
            r.grad = None

    with torch.no_grad():
        tf_f, tf_t = timeit(fused_f), timeit(torch_f)
    tb_f, tb_t = timeit(lambda: fb(fused_f)), timeit(lambda: fb(torch_f))
    pf = 4 if res else 3
    pb = pf + (8 if res else 5)
    print(f"N={N} C={C:4d} {H}x{W} res={int(res)} ({mb:7.1f} MB): fwd fused {tf_f:7.1f} us ({pf*mb/tf_f:4.2f} TB/s) torch {tf_t:7.1f} us"
          f" | fwd+bwd fused {tb_f:7.1f} us ({pb*mb/tb_f:4.2f} TB/s) torch {tb_t:7.1f} us")
