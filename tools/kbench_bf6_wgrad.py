"""dev tool: weight-gradient time of the bf16x6 GEMM only, on the step's 1x1 shapes (one line; A/B runs through DBEV_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distill_bev_amd import _lib as L
dev = torch.device("cuda:0")
SHAPES = [(48, 128, 512, 32, 88), (48, 512, 128, 32, 88), (48, 256, 1024, 16, 44), (48, 1024, 256, 16, 44), (48, 512, 2048, 8, 22),
          (48, 2048, 512, 8, 22), (48, 1024, 512, 16, 44), (8, 256, 256, 128, 128), (8, 512, 512, 64, 64), (48, 256, 128, 64, 176)]
def tm(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
out = []
for (n, ci, co, h, w) in SHAPES:
    M = n * h * w
    x = torch.relu(torch.randn((n, ci, h, w), device=dev)).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((n, co, h, w), device=dev).contiguous(memory_format=torch.channels_last)
    gw = torch.empty((co, ci), device=dev)
    nb = int(L.call("dbev_gemm_bf16x6_backward_weight_workspace_bytes", M, ci, co, ci))
    ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
    out.append(tm(lambda: L.call("dbev_gemm_bf16x6_backward_weight", L.ptr(x), L.ptr(gy), L.ptr(gw), M, ci, co, ci, L.ptr(ws), nb, L.stream_ptr(dev))))
print(os.path.basename(L.LIB_PATH), " ".join(f"{t:6.1f}" for t in out), " sum %.1f" % sum(out))
