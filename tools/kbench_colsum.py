"""dev tool: channel_sum (csrc/colsum.hip) vs ATen's t.sum((0, 2, 3)) on the step's bias-gradient shapes, us per call"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distill_bev_amd.colsum import channel_sum
dev = torch.device("cuda:0")
def tm(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for shp in [(48, 59, 16, 44), (48, 27, 16, 44), (48, 256, 16, 44), (48, 512, 16, 44), (8, 256, 128, 128), (8, 384, 128, 128), (8, 512, 64, 64),
            (8, 128, 64, 64), (48, 512, 8, 22), (48, 256, 64, 176)]:
    t = torch.randn(shp, device=dev).contiguous(memory_format=torch.channels_last)
    mb = t.numel() * 4 / 1e6
    a, b = tm(lambda: channel_sum(t)), tm(lambda: t.sum((0, 2, 3)))
    print(f"{str(shp):24s} {mb:8.1f} MB  colsum {a:7.1f} us ({mb / a / 1e3 * 1e3:6.2f} GB/ms)   aten {b:7.1f} us")
