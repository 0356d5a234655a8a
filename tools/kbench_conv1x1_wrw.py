"""Weight gradient of the 1x1 convolutions of ResNet-50 at the training step's shapes (channels-last, fp32):
MIOpen's convolution_backward (split-K implicit GEMM + zeroing launches) vs one GEMM dW = dy^T x on the [pixels, C] views."""
import torch

dev = torch.device("cuda:0")


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


shapes = [(96, 64, 64, 176, 256), (96, 256, 64, 176, 64), (96, 256, 64, 176, 128), (96, 128, 32, 88, 512), (96, 512, 32, 88, 128),
          (96, 512, 32, 88, 256), (96, 256, 16, 44, 1024), (96, 1024, 16, 44, 256), (96, 1024, 16, 44, 512), (96, 512, 8, 22, 2048),
          (96, 2048, 8, 22, 512)]
tot = [0.0, 0.0, 0.0, 0.0]
for (N, Ci, H, W, Co) in shapes:
    x = torch.randn(N, Ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, Co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, Ci, 1, 1, device=dev).contiguous(memory_format=torch.channels_last)
    x2, dy2 = x.permute(0, 2, 3, 1).reshape(-1, Ci), dy.permute(0, 2, 3, 1).reshape(-1, Co)

    def mi_w():
        return torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]

    def mm_w():
        return dy2.t() @ x2

    def mi_x():
        return torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]

    def mm_x():
        return dy2 @ w.view(Co, Ci)
    a, b, c, d = timeit(mi_w), timeit(mm_w), timeit(mi_x), timeit(mm_x)
    err = float((mi_w().view(Co, Ci) - mm_w()).abs().max() / mm_w().abs().max())
    tot = [tot[0] + a, tot[1] + b, tot[2] + c, tot[3] + d]
    print(f"{Ci:5d}->{Co:5d} @ {N}x{H}x{W}: dW MIOpen {a:.3f} ms  GEMM {b:.3f} ms | dx MIOpen {c:.3f} ms  GEMM {d:.3f} ms   (dW rel diff {err:.1e})")
print("sum: dW MIOpen %.2f  GEMM %.2f | dx MIOpen %.2f  GEMM %.2f" % tuple(tot))
