#!/usr/bin/env python3
"""csrc/stem.hip at the bench shape (48 x 3 x 256 x 704): forward (+statistics) and weight gradient vs the library, HIP-event timed.
usage: kbench_stem.py [N H W]   (DBEV_HIP_LIB=<variant .so> for tools/build_variant.sh builds)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from distill_bev_amd import stem

N, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (48, 256, 704)
dev = torch.device("cuda:0")
x = torch.randn((N, 3, H, W), device=dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn((64, 3, 7, 7), device=dev) * 0.1).contiguous(memory_format=torch.channels_last).requires_grad_(True)
Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
gz = torch.randn((N, 64, Ho, Wo), device=dev).contiguous(memory_format=torch.channels_last)
flops = 2.0 * N * Ho * Wo * 64 * 147


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


rows = [("stem fwd + stats", lambda: stem.stem_conv_stats(x, w.detach())),
        ("stem fwd", lambda: stem.stem_conv(x, w.detach())),
        ("library fwd", lambda: F.conv2d(x, w.detach(), None, 2, 3)),
        ("stem wgrad", lambda: stem._StemConv.backward(type("C", (), {"saved_tensors": (x, w), "needs_input_grad": (False, True, False)})(), gz)),
        ("library wgrad", lambda: torch.ops.aten.convolution_backward(gz, x, w.detach(), None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False]))]
for name, fn in rows:
    ms = timed(fn)
    print(f"{name:18s} {ms * 1e3:8.1f} us   {flops / ms / 1e9:6.1f} TFLOP/s")
