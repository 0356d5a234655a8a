#!/usr/bin/env python3
"""Per-layer roofline of the ResNet-50 convolutions of the step (MIOpen fp32 NHWC): forward, data gradient and weight gradient
of every distinct layer shape at the step's batch (48 images of 256 x 704 per frame), against
max(FLOPs / 157.3 TFLOP/s, algorithmic bytes / 8 TB/s).  Answers: is the image backbone MFMA-bound or HBM-bound, layer by layer,
and how far is the library from either bound.

    python tools/conv_roofline.py [N]
"""
import sys

import torch

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
PEAK_TF, PEAK_BW = 157.3, 8.0e12


def timeit(fn, n=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


def layers():
    """(name, Cin, Cout, k, stride, Hin, Win, count per forward)"""
    out = [("stem 7x7/2", 3, 64, 7, 2, 256, 704, 1)]
    H, W, cin = 64, 176, 64
    for si, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]):
        for b in range(blocks):
            s = stride if b == 0 else 1
            tag = f"l{si + 1}.{b}"
            out.append((tag + " 1x1 reduce", cin, planes, 1, 1, H, W, 1))
            out.append((tag + " 3x3", planes, planes, 3, s, H, W, 1))       # torchvision-style: stride on the 3x3
            Ho, Wo = H // s, W // s
            out.append((tag + " 1x1 expand", planes, planes * 4, 1, 1, Ho, Wo, 1))
            if b == 0:
                out.append((tag + " downsample", cin, planes * 4, 1, s, H, W, 1))
            cin, H, W = planes * 4, Ho, Wo
    # merge identical shapes
    merged = {}
    for name, ci, co, k, s, h, w, c in out:
        key = (ci, co, k, s, h, w)
        if key in merged:
            merged[key][1] += c
        else:
            merged[key] = [name, c]
    return [(v[0], *k, v[1]) for k, v in merged.items()]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wrw": [0.0, 0.0]}
    print(f"{'layer':18s} {'shape':28s} cnt   GF    MB  | {'fwd ms':>7s} {'TF':>5s} {'TB/s':>5s} {'%roof':>5s} | {'dgrad':>7s} {'%roof':>5s} | {'wrw':>7s} {'%roof':>5s}")
    for name, ci, co, k, s, h, w, cnt in layers():
        x = torch.randn((N, ci, h, w), device=dev).contiguous(memory_format=torch.channels_last)
        wt = (torch.randn((co, ci, k, k), device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        pad = k // 2
        y = torch.nn.functional.conv2d(x, wt, None, s, pad)
        gy = torch.randn_like(y)
        ho, wo = y.shape[2:]
        flops = 2.0 * N * ho * wo * co * ci * k * k
        bytes_ = 4.0 * (x.numel() + y.numel() + wt.numel())
        roof = max(flops / (PEAK_TF * 1e12), bytes_ / PEAK_BW)
        args = (gy, x, wt, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1)
        tf = timeit(lambda: torch.nn.functional.conv2d(x, wt, None, s, pad))
        td = timeit(lambda: torch.ops.aten.convolution_backward(*args, [True, False, False])) if ci > 3 else float("nan")
        tw = timeit(lambda: torch.ops.aten.convolution_backward(*args, [False, True, False]))
        print(f"{name:18s} {f'{ci}->{co} k{k} s{s} {h}x{w}':28s} {cnt:3d} {flops / 1e9:5.0f} {bytes_ / 1e6:5.0f}  | {tf * 1e3:7.3f} {flops / tf / 1e12:5.0f} {bytes_ / tf / 1e12:5.2f} {100 * roof / tf:5.0f} |"
              f" {td * 1e3:7.3f} {100 * roof / td:5.0f} | {tw * 1e3:7.3f} {100 * roof / tw:5.0f}")
        for kk, t in (("fwd", tf), ("dgrad", td), ("wrw", tw)):
            if t == t:
                tot[kk][0] += t * cnt; tot[kk][1] += roof * cnt
        del x, wt, y, gy
    for kk, (t, r) in tot.items():
        print(f"total {kk:6s} {t * 1e3:8.2f} ms   roofline {r * 1e3:8.2f} ms   = {100 * r / t:.0f} % of the bound")


if __name__ == "__main__":
    main()
