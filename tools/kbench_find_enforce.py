#!/usr/bin/env python3
"""Does MIOpen's parameter search (MIOPEN_FIND_ENFORCE=3) beat its plain exhaustive find on the step's heaviest shape?  (dev tool)
    python tools/kbench_find_enforce.py            # run once with and once without MIOPEN_FIND_ENFORCE=3 in the environment"""
import os, time, sys
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


for (N, C, K, H, W, k) in [(48, 256, 256, 16, 44, 3), (8, 64, 2048, 128, 128, 3)]:
    x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((K, C, k, k), device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = F.conv2d(x, w, None, 1, k // 2)
    gy = torch.randn_like(y)
    fl = 2.0 * N * H * W * C * K * k * k
    args = (gy, x, w, None, [1, 1], [k // 2] * 2, [1, 1], False, [0, 0], 1)
    t0 = time.time()
    tf = timeit(lambda: F.conv2d(x, w, None, 1, k // 2))
    t1 = time.time()
    td = timeit(lambda: torch.ops.aten.convolution_backward(*args, [True, False, False]))
    t2 = time.time()
    tw = timeit(lambda: torch.ops.aten.convolution_backward(*args, [False, True, False]))
    t3 = time.time()
    print(f"ENFORCE={os.environ.get('MIOPEN_FIND_ENFORCE', '-')} N{N} {C}->{K} k{k} {H}x{W}: fwd {tf:.3f} ms {fl / tf / 1e9:4.0f} TF (search {t1 - t0:.0f} s) | "
          f"dgrad {td:.3f} ms {fl / td / 1e9:4.0f} TF ({t2 - t1:.0f} s) | wrw {tw:.3f} ms {fl / tw / 1e9:4.0f} TF ({t3 - t2:.0f} s)", flush=True)
