#!/usr/bin/env python3
"""Frozen-teacher layers: conv -> eval BatchNorm -> ReLU as (a) MIOpen conv + dbev_bn_act_infer (now), (b) BatchNorm folded into the
weights + torch.miopen_convolution_relu (MIOpen's fused conv-bias-activation), at the SECOND / FPN shapes of the step (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
from distill_bev_amd.bn_act import bn_act
dev = torch.device("cuda:0")


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for (N, Ci, Co, H, W, st) in [(8, 64, 64, 512, 512, 2), (8, 64, 64, 256, 256, 1), (8, 64, 128, 256, 256, 2), (8, 128, 128, 128, 128, 1),
                              (8, 128, 256, 128, 128, 2), (8, 256, 256, 64, 64, 1)]:
    x = torch.randn((N, Ci, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    conv = nn.Conv2d(Ci, Co, 3, st, 1, bias=False).to(dev).to(memory_format=torch.channels_last)
    bn = nn.BatchNorm2d(Co, eps=1e-3).to(dev).eval()
    bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_(0, 0.2)
    sc = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    wf = (conv.weight * sc.view(-1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
    bf = (bn.bias - bn.running_mean * sc).contiguous()
    with torch.no_grad():
        ref = bn_act(conv(x), bn, None, True)
        ta = timeit(lambda: bn_act(conv(x), bn, None, True))
        try:
            out = torch.miopen_convolution_relu(x, wf, bf, [st, st], [1, 1], [1, 1], 1)
            err = float((out - ref).abs().max() / ref.abs().max())
            tb = timeit(lambda: torch.miopen_convolution_relu(x, wf, bf, [st, st], [1, 1], [1, 1], 1))
            cl = out.is_contiguous(memory_format=torch.channels_last)
        except Exception as ex:
            err, tb, cl = str(ex)[:80], float("nan"), None
        tc = timeit(lambda: conv(x))
    print(f"{N}x{Ci}->{Co} {H}x{W}/s{st}: conv+bn_infer {ta:7.1f} us (conv alone {tc:7.1f}) | fused conv-bias-relu {tb:7.1f} us err {err} channels_last={cl}", flush=True)
