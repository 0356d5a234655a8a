"""1x1-convolution GEMM kernels vs MIOpen on the step's 1x1 layer shapes: forward, data gradient, weight gradient.
python tools/kbench_gemm1x1.py [iters]"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distill_bev_amd import gemm1x1 as G
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
dev = torch.device("cuda:0")
it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SH = [(48, 64, 256, 64, 176), (48, 256, 64, 64, 176), (48, 128, 512, 32, 88), (48, 512, 128, 32, 88), (48, 256, 1024, 16, 44),
      (48, 1024, 256, 16, 44), (48, 512, 2048, 8, 22), (48, 2048, 512, 8, 22), (48, 1024, 512, 16, 44), (48, 2304, 256, 16, 44),
      (8, 256, 256, 128, 128), (8, 512, 512, 64, 64), (48, 256, 128, 64, 176), (48, 512, 256, 32, 88)]


def t(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


print(f"{'shape':30s} {'GF':>6s} | fwd us (TF) vs miopen | dgrad us vs miopen | wgrad us vs miopen")
for N, C, Co, H, W in SH:
    x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Co, C, 1, 1), device=dev) / C ** 0.5).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((N, Co, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    assert G.eligible(x, w)
    gf = 2.0 * N * H * W * C * Co / 1e9
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    tf = t(lambda: G._Conv1x1.apply(x, w, False)); tfs = t(lambda: G._Conv1x1.apply(x, w, True)); tm = t(lambda: F.conv2d(x, w))
    cb = torch.ops.aten.convolution_backward
    from distill_bev_amd import _lib as L
    M = N * H * W
    wt = w.reshape(Co, C).t().contiguous()
    td = t(lambda: G._gemm(gy, M, Co, C, wt, False, dev, (N, C, H, W))); tdm = t(lambda: cb(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]))
    nbytes = int(L.call("dbev_gemm1x1_backward_weight_workspace_bytes", M, C, Co, C))
    g2 = torch.empty((Co, C), device=dev); ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    tw = t(lambda: L.call("dbev_gemm1x1_backward_weight", L.ptr(x), L.ptr(gy), L.ptr(g2), M, C, Co, C, L.ptr(ws), nbytes, L.stream_ptr(dev)))
    twm = t(lambda: cb(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
    print(f"{str((N, C, Co, H, W)):30s} {gf:6.1f} | {tf:6.1f} ({gf / tf * 1e3:5.1f}) +stats {tfs:6.1f} vs {tm:6.1f} {tm / tf:4.2f}x | {td:6.1f} vs {tdm:6.1f} {tdm / td:4.2f}x | "
          f"{tw:6.1f} vs {twm:6.1f} {twm / tw:4.2f}x", flush=True)
