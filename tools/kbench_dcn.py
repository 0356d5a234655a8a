#!/usr/bin/env python3
"""DCNv2 at the depth-head shape (N=48, C=256, 16x44): HIP sampling kernels vs the grid_sample restatement (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distill_bev_amd import _lib as L
from distill_bev_amd.dcn import ModulatedDeformConv2dPack
from oracle.dcn import pack_forward

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = ModulatedDeformConv2dPack(256, 256, 3, padding=1).to(dev).to(memory_format=torch.channels_last)
torch.nn.init.normal_(m.conv_offset.weight, std=0.02)
x = torch.randn(48, 256, 16, 44, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)


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


def step(f):
    y = f(x)
    y.backward(torch.ones_like(y))
    x.grad = None; m.zero_grad(set_to_none=True)


for k in ("dbev_dcnv2_im2col", "dbev_dcnv2_col2im"):
    L.enable_timing(k)
print(f"hip  fwd+bwd: {timeit(lambda: step(m)):8.1f} us")
for k in ("dbev_dcnv2_im2col", "dbev_dcnv2_col2im"):
    ms = L.timing_ms(k)[5:]
    print(f"   {k:22s} avg {1e3 * sum(ms) / len(ms):8.1f} us")
print(f"torch fwd+bwd: {timeit(lambda: step(lambda t: pack_forward(m, t))):8.1f} us")
with torch.no_grad():
    print(f"hip  fwd only: {timeit(lambda: m(x)):8.1f} us    torch fwd only: {timeit(lambda: pack_forward(m, x)):8.1f} us")
