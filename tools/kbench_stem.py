#!/usr/bin/env python3
"""The 7x7 / stride 2 stem convolution (Cin = 3): NHWC vs NCHW, and the input padded to 4 channels with a zero weight channel
(same arithmetic), forward and weight gradient, at the two recipes' shapes (dev tool)."""
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
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
    return s.elapsed_time(e) / n


for (N, H, W) in [(48, 256, 704), (6, 928, 1600), (6, 480, 800)]:
    x3 = torch.randn((N, 3, H, W), device=dev)
    w3 = torch.randn((64, 3, 7, 7), device=dev) * 0.05
    ref = F.conv2d(x3, w3, None, 2, 3)
    gy = torch.randn_like(ref)
    for name, cin, cl in (("NCHW c3", 3, False), ("NHWC c3", 3, True), ("NHWC c4", 4, True), ("NCHW c4", 4, False), ("NHWC c8", 8, True)):
        x = x3 if cin == 3 else torch.cat([x3, x3.new_zeros((N, cin - 3, H, W))], 1)
        w = w3 if cin == 3 else torch.cat([w3, w3.new_zeros((64, cin - 3, 7, 7))], 1)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last); w = w.contiguous(memory_format=torch.channels_last)
            g = gy.contiguous(memory_format=torch.channels_last)
        else:
            x = x.contiguous(); w = w.contiguous(); g = gy.contiguous()
        y = F.conv2d(x, w, None, 2, 3)
        err = float((y - ref).abs().max() / ref.abs().max())
        tf = timeit(lambda: F.conv2d(x, w, None, 2, 3))
        tw = timeit(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False]))
        fl = 2.0 * N * (H // 2) * (W // 2) * 64 * 147
        print(f"N{N} {H}x{W} {name}: fwd {tf:.3f} ms ({fl / tf / 1e9:.0f} TF)  wrw {tw:.3f} ms ({fl / tw / 1e9:.0f} TF)  max rel diff vs NCHW c3 {err:.1e}")
