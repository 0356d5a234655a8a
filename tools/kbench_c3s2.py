#!/usr/bin/env python3
"""3x3 / stride-2 convolution forward at the image backbone's three shapes: implicit bf16x6 GEMM (+ statistics) vs the library (dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from distill_bev_amd import gemm_bf6 as G
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
dev = torch.device("cuda:0")


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


for (N, C, H, W) in [(48, 128, 64, 176), (48, 256, 32, 88), (48, 512, 16, 44), (8, 128, 128, 128)]:
    x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((C, C, 3, 3), device=dev) / (3 * C ** 0.5)).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * N * (H // 2) * (W // 2) * C * C * 9
    ok = G.eligible_c3s2(x, w)
    t_lib = timed(lambda: F.conv2d(x, w, None, 2, 1))
    if ok:
        t = timed(lambda: G.product_c3s2(x, w)); ts = timed(lambda: G.product_c3s2(x, w, True))
        print(f"{(N, C, H, W)}: bf16x6 {t * 1e3:7.1f} us {fl / t / 1e9:6.1f} TF | + stats {ts * 1e3:7.1f} us | library {t_lib * 1e3:7.1f} us {fl / t_lib / 1e9:6.1f} TF")
    else:
        print(f"{(N, C, H, W)}: not eligible | library {t_lib * 1e3:7.1f} us {fl / t_lib / 1e9:6.1f} TF")
