#!/usr/bin/env python3
"""1x1 convolution (MIOpen NHWC igemm) vs the same contraction as a plain GEMM (hipBLASLt via F.linear) at the
ResNet-50 shapes of the step, fwd and fwd+bwd, fp32 (dev tool)."""
import os, sys
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")


def timeit(fn, n=10, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for (N, H, W, Ci, Co) in [(48, 64, 176, 64, 64), (48, 64, 176, 64, 256), (48, 64, 176, 256, 64), (48, 32, 88, 512, 128),
                          (48, 32, 88, 128, 512), (48, 16, 44, 1024, 256), (48, 16, 44, 256, 1024), (48, 8, 22, 2048, 512),
                          (48, 8, 22, 512, 2048), (8, 128, 128, 256, 384)]:
    x = torch.randn((N, Ci, H, W), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn((Co, Ci, 1, 1), device=dev) * 0.05).requires_grad_(True)
    g = torch.randn((N, Co, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    x2 = x.detach().permute(0, 2, 3, 1).reshape(-1, Ci).requires_grad_(True)
    w2 = w.detach().reshape(Co, Ci).requires_grad_(True)
    g2 = g.permute(0, 2, 3, 1).reshape(-1, Co)

    def conv_fb():
        y = F.conv2d(x, w); y.backward(g); x.grad = None; w.grad = None

    def lin_fb():
        y = F.linear(x2, w2); y.backward(g2); x2.grad = None; w2.grad = None

    with torch.no_grad():
        tcf = timeit(lambda: F.conv2d(x, w)); tlf = timeit(lambda: F.linear(x2, w2))
    tcb = timeit(conv_fb); tlb = timeit(lin_fb)
    fl = 2 * N * H * W * Ci * Co / 1e12
    print(f"M={N*H*W:7d} {Ci:4d}->{Co:4d}: fwd conv {tcf:7.1f} us ({fl/tcf*1e6:5.1f} TF/s) gemm {tlf:7.1f} us ({fl/tlf*1e6:5.1f} TF/s) | "
          f"fwd+bwd conv {tcb:7.1f} us gemm {tlb:7.1f} us")
