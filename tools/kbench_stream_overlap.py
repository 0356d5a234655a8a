"""Does an MFMA-bound MIOpen weight-gradient kernel overlap with an HBM-bound normalisation pass when the two are issued on
different HIP streams?  (design question for running the wrw GEMMs of the backward pass on a side stream)

    python tools/kbench_stream_overlap.py
"""
import time

import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    side = torch.cuda.Stream()
    for (N, C, K, H, W, ks) in [(96, 64, 64, 64, 176, 3), (96, 256, 64, 64, 176, 1), (96, 128, 128, 32, 88, 3), (96, 512, 2048, 8, 22, 1)]:
        x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(K, C, ks, ks, device=dev).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(N, K, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        a = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        b = torch.empty_like(a)

        def wrw():
            return torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [ks // 2] * 2, [1, 1], False, [0, 0], 1, [False, True, False])[1]

        def dgrad():
            return torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [ks // 2] * 2, [1, 1], False, [0, 0], 1, [True, False, False])[0]

        def stream_op():                 # HBM-bound: 2 reads + 1 write of the activation
            torch.add(a, x, out=b)

        def seq(f):
            def run():
                f(); stream_op(); stream_op()
            return run

        def par(f):
            def run():
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    f()
                stream_op(); stream_op()
                torch.cuda.current_stream().wait_stream(side)
            return run

        t_w, t_d, t_s = timeit(wrw), timeit(dgrad), timeit(lambda: (stream_op(), stream_op()))
        print(f"N{N} C{C} K{K} {H}x{W} k{ks}: wrw {t_w:.3f} dgrad {t_d:.3f} 2xadd {t_s:.3f} | wrw seq {timeit(seq(wrw)):.3f} par {timeit(par(wrw)):.3f}"
              f" | dgrad seq {timeit(seq(dgrad)):.3f} par {timeit(par(dgrad)):.3f}"
              f" | wrw||dgrad seq {timeit(lambda: (wrw(), dgrad())):.3f} par {timeit(lambda: (par(wrw)(), None) if False else _both(side, wrw, dgrad)):.3f}")


def _both(side, f, g):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        f()
    g()
    torch.cuda.current_stream().wait_stream(side)


if __name__ == "__main__":
    main()
