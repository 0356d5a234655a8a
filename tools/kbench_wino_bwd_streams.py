"""data gradient and weight gradient of one 3x3 layer: back to back on one stream vs on two streams (the weight gradient's single round of
workgroups fills the CUs the data gradient's last, partly filled round leaves idle, and the other way round).  dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distill_bev_amd import wino
dev = torch.device("cuda:0")
side = torch.cuda.Stream(dev)
for N, C, Co, H, W in [(48, 256, 256, 16, 44), (8, 512, 512, 64, 64), (48, 64, 64, 64, 176), (48, 128, 128, 32, 88), (8, 64, 64, 128, 128),
                       (8, 128, 128, 128, 128), (8, 256, 256, 64, 64), (48, 512, 512, 8, 22)]:
    x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Co, C, 3, 3), device=dev) / (3 * C ** 0.5)).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((N, Co, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    Ud = wino.pack_filters(w, True, gy.shape)

    def seq():
        wino.conv_packed(gy, Ud, C)
        wino.weight_gradient(x, gy, w)

    def par():
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            wino.weight_gradient(x, gy, w)
        wino.conv_packed(gy, Ud, C)
        main.wait_stream(side)

    res = []
    for fn in (seq, par, seq, par):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record()
        for _ in range(20):
            fn()
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 20 * 1e3)
    print("%-28s  one stream %7.1f %7.1f us   two streams %7.1f %7.1f us   %.3f" % ((N, C, Co, H, W), res[0], res[2], res[1], res[3], min(res[1], res[3]) / min(res[0], res[2])), flush=True)
