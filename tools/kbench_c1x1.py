#!/usr/bin/env python3
"""dbev_conv1x1_forward (fp32-MFMA 1x1 convolution with the BatchNorm statistics of its output in the epilogue) vs MIOpen's NHWC
igemm for the same layer + the separate statistics pass it makes unnecessary, at the ResNet-50 shapes of the step (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
from distill_bev_amd import _lib as L
dev = torch.device("cuda:0")


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


def ours(x2, w2, y2, part):
    M, K = x2.shape
    N = w2.shape[0]
    L.call("dbev_conv1x1_forward", L.ptr(x2), L.ptr(w2), L.ptr(y2), L.ptr(part), M, K, N, K, L.stream_ptr(dev))


shapes = [(48, 64, 176, 64, 64), (48, 64, 176, 64, 256), (48, 64, 176, 256, 64), (48, 64, 176, 256, 128), (48, 32, 88, 512, 128),
          (48, 32, 88, 128, 512), (48, 16, 44, 1024, 256), (48, 16, 44, 256, 1024), (48, 8, 22, 2048, 512), (48, 8, 22, 512, 2048)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (N, H, W, Ci, Co) in shapes:
    x = torch.randn((N, Ci, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Co, Ci, 1, 1), device=dev) * 0.05)
    M = N * H * W
    x2 = x.permute(0, 2, 3, 1).reshape(M, Ci)
    w2 = w.reshape(Co, Ci).contiguous()
    y2 = torch.empty((M, Co), device=dev)
    rows = int(L.call("dbev_conv1x1_stats_rows", M, Ci, Co))
    part = torch.empty((rows, 2, Co), device=dev)
    with torch.no_grad():
        yref = F.conv2d(x, w)
        ours(x2, w2, y2, part)
        torch.cuda.synchronize()
        err = float((y2 - yref.permute(0, 2, 3, 1).reshape(M, Co)).abs().max() / yref.abs().max())
        s = part.double().sum(0)
        serr = float((s[0] - y2.double().sum(0)).abs().max() / y2.double().sum(0).abs().max())
        qerr = float((s[1] - y2.double().square().sum(0)).abs().max() / y2.double().square().sum(0).abs().max())
        tm = timeit(lambda: F.conv2d(x, w))
        to = timeit(lambda: ours(x2, w2, y2, part))
        ton = timeit(lambda: ours(x2, w2, y2, None))
        bn = torch.nn.BatchNorm2d(Co).to(dev).train()
        from distill_bev_amd import bn_act as BA
        L.kernel_timing(["bn_stats"])
        L.kernel_timing_read()
        for _ in range(10):
            BA.bn_act(yref, bn, None, True)
        rec = L.kernel_timing_read().get("bn_stats", [])
        L.kernel_timing(False)
        ts = sum(r[0] for r in rec[3:]) / max(len(rec) - 3, 1) * 1e3
    fl = 2 * M * Ci * Co / 1e12
    by = 4 * M * (Ci + Co) / 1e9
    print(f"M={M:7d} {Ci:4d}->{Co:4d} rows={rows:3d}: MIOpen {tm:7.1f} us ({fl/tm*1e6:5.1f} TF {by/tm*1e3:5.2f} TB/s) + bn_stats {ts:6.1f} us | "
          f"ours+stats {to:7.1f} us ({fl/to*1e6:5.1f} TF {by/to*1e3:5.2f} TB/s) no-stats {ton:7.1f} us | err {err:.1e} sum {serr:.1e} sq {qerr:.1e}", flush=True)
