#!/usr/bin/env python3
"""MIOpen fp32 NHWC 3x3 / 1x1 convolutions at the BEVFormer image sizes: how the library's rate depends on N*H*W divisibility
(58 x 100 maps of the 928 x 1600 recipe run at 45 TFLOP/s, the 16 x 44 maps of the BEVDepth recipe at 110) (dev tool)."""
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")


def timeit(fn, n=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for (N, C, K, H, W, k) in [(18, 256, 256, 58, 100, 3), (6, 256, 256, 58, 100, 3), (18, 256, 256, 58, 104, 3), (18, 256, 256, 60, 100, 3),
                           (18, 256, 256, 64, 100, 3), (18, 256, 256, 56, 96, 3), (16, 256, 256, 58, 100, 3), (18, 256, 256, 64, 104, 3),
                           (18, 256, 1024, 58, 100, 1), (18, 1024, 256, 58, 100, 1), (18, 256, 1024, 64, 104, 1), (18, 64, 64, 232, 400, 3),
                           (18, 128, 128, 116, 200, 3), (18, 512, 512, 29, 50, 3), (18, 512, 512, 32, 52, 3)]:
    x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((K, C, k, k), device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = F.conv2d(x, w, None, 1, k // 2)
    gy = torch.randn_like(y)
    fl = 2.0 * N * H * W * C * K * k * k
    args = (gy, x, w, None, [1, 1], [k // 2] * 2, [1, 1], False, [0, 0], 1)
    tf = timeit(lambda: F.conv2d(x, w, None, 1, k // 2))
    td = timeit(lambda: torch.ops.aten.convolution_backward(*args, [True, False, False]))
    tw = timeit(lambda: torch.ops.aten.convolution_backward(*args, [False, True, False]))
    print(f"N{N} {C}->{K} k{k} {H}x{W} (NHW = {N * H * W} = {N * H * W / 128:.2f} x 128): fwd {tf:.3f} ms {fl / tf / 1e9:4.0f} TF | dgrad {td:.3f} ms {fl / td / 1e9:4.0f} TF | wrw {tw:.3f} ms {fl / tw / 1e9:4.0f} TF")
