"""Winograd 3x3 kernel vs MIOpen on the step's 3x3 stride-1 layer shapes (forward; the data gradient is the same kernel).
python tools/kbench_wino.py [iters]"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from distill_bev_amd import wino
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
dev = torch.device("cuda:0")
it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SH = [(48, 256, 256, 16, 44), (48, 128, 128, 32, 88), (48, 64, 64, 64, 176), (48, 512, 512, 8, 22), (8, 64, 64, 128, 128),
      (8, 512, 256, 128, 128), (8, 128, 128, 128, 128), (8, 256, 256, 64, 64), (8, 512, 512, 64, 64), (8, 640, 512, 64, 64),
      (8, 64, 2304, 128, 128), (8, 64, 64, 256, 256), (8, 256, 256, 32, 32), (8, 512, 512, 16, 16), (48, 512, 512, 16, 44)]


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


print(f"{'shape':34s} {'GF':>7s} | {'wino us':>8s} {'eff TF':>7s} | {'miopen us':>9s} {'TF':>6s} | speedup   err/direct")
for N, C, Co, H, W in SH:
    x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Co, C, 3, 3), device=dev) / (3 * C ** 0.5)).contiguous(memory_format=torch.channels_last)
    assert wino.eligible(x, w), (N, C, Co, H, W)
    U = wino.pack_filters(w)
    gf = 2.0 * N * H * W * C * Co * 9 / 1e9
    tw = t(lambda: wino.conv_packed(x, U, Co))
    tm = t(lambda: F.conv2d(x, w, None, 1, 1))
    tp = t(lambda: wino.pack_filters(w))
    gy = torch.randn((N, Co, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    if C % 64 == 0:
        tgw = t(lambda: wino.weight_gradient(x, gy, w))
        tgm = t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
        wg = f" | wgrad {tgw:7.1f} vs {tgm:7.1f} us {tgm / tgw:4.2f}x"
    else:
        wg = ""
    y, ym = wino.conv_packed(x, U, Co), F.conv2d(x, w, None, 1, 1)
    ref = F.conv2d(x[:2].double(), w.double(), None, 1, 1)
    e1 = float((y[:2].double() - ref).abs().max() / ref.abs().max()); e2 = float((ym[:2].double() - ref).abs().max() / ref.abs().max())
    print(f"{str((N, C, Co, H, W)):34s} {gf:7.1f} | {tw:8.1f} {gf / tw * 1e3:7.1f} | {tm:9.1f} {gf / tm * 1e3:6.1f} | {tm / tw:5.2f}x  pack {tp:5.1f} us  {e1:.1e}/{e2:.1e}{wg}",
          flush=True)
