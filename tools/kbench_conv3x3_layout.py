"""3x3 convolutions of ResNet-50 at the training step's shapes: NCHW (MIOpen may pick its fp32 Winograd kernels) vs channels-last
(implicit GEMM), forward and forward + backward."""
import torch
import torch.nn.functional as F

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


for (N, C, H, W) in ((96, 64, 64, 176), (96, 128, 32, 88), (96, 256, 16, 44), (96, 512, 8, 22)):
    row = []
    for fmt in (torch.contiguous_format, torch.channels_last):
        x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=fmt).requires_grad_(True)
        w = torch.randn(C, C, 3, 3, device=dev).contiguous(memory_format=fmt).requires_grad_(True)

        def fwd():
            with torch.no_grad():
                return F.conv2d(x, w, padding=1)

        def fb():
            y = F.conv2d(x, w, padding=1)
            torch.autograd.grad(y, [x, w], torch.ones_like(y))
        row += [timeit(fwd), timeit(fb)]
    gf = 2.0 * N * H * W * C * C * 9 / 1e9
    print(f"{N}x{C}x{H}x{W}: NCHW fwd {row[0]:.3f} ms ({gf / row[0]:.0f} TF-equiv) f+b {row[1]:.3f} | NHWC fwd {row[2]:.3f} ms ({gf / row[2]:.0f} TF) f+b {row[3]:.3f}")
