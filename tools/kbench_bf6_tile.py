"""dev tool: 64- vs 128-column tiles of the bf16x6 forward kernel on the layers with few row blocks.  python tools/kbench_bf6_tile.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distill_bev_amd import _lib as L
dev = torch.device("cuda:0")
def tm(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (n, ci, co, h, w) in [(48, 64, 256, 64, 176), (48, 2048, 512, 8, 22), (48, 512, 2048, 8, 22), (48, 1024, 256, 16, 44), (48, 512, 512, 8, 22), (8, 256, 256, 32, 32), (8, 512, 256, 16, 16)]:
    M = n * h * w
    x = torch.randn((n, ci, h, w), device=dev).contiguous(memory_format=torch.channels_last)
    w2 = torch.randn((co, ci), device=dev) / ci ** 0.5
    y = torch.empty((n, co, h, w), device=dev).contiguous(memory_format=torch.channels_last)
    res = []
    for tn in (64, 128):
        nb = int(L.call("dbev_gemm_bf16x6_packed_bytes", co, ci))
        p = torch.empty((nb,), dtype=torch.uint8, device=dev)
        L.call("dbev_gemm_bf16x6_pack", L.ptr(w2), w2.stride(0), w2.stride(1), co, ci, tn, L.ptr(p), L.stream_ptr(dev))
        t = tm(lambda: L.call("dbev_gemm_bf16x6_forward", L.ptr(x), L.ptr(p), L.ptr(y), M, ci, co, ci, tn, L.stream_ptr(dev)))
        res.append(t)
    gf = 2.0 * M * ci * co / 1e9
    print(f"{(n, ci, co, h, w)}  rows/128 = {M // 128}  tiles64 = {M // 128 * (co // 64)}  tiles128 = {M // 128 * (co // 128)}   64: {res[0]:7.1f} us {gf / res[0] * 1e3:6.1f} TF   128: {res[1]:7.1f} us {gf / res[1] * 1e3:6.1f} TF")
