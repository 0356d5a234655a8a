"""How much would batching the 36 first branch convolutions (64 -> 64, 3x3) of a CenterHead into one 64 -> 2304 convolution
buy?  (bs 8, 128 x 128, fp32, channels-last; MIOpen picks the kernels.)"""
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")
x = torch.randn(8, 64, 128, 128, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
ws = [torch.randn(64, 64, 3, 3, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) for _ in range(36)]
wall = torch.cat([w.detach() for w in ws]).contiguous(memory_format=torch.channels_last).requires_grad_(True)


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def sep_fwd():
    with torch.no_grad():
        return [F.conv2d(x, w, padding=1) for w in ws]


def bat_fwd():
    with torch.no_grad():
        return F.conv2d(x, wall, padding=1)


def sep_fb():
    ys = [F.conv2d(x, w, padding=1) for w in ws]
    torch.autograd.grad([y.sum() for y in ys], [x] + ws)


def bat_fb():
    y = F.conv2d(x, wall, padding=1)
    torch.autograd.grad(y.sum(), [x, wall])


print("separate fwd %.3f ms   batched fwd %.3f ms" % (timeit(sep_fwd), timeit(bat_fwd)))
print("separate fwd+bwd %.3f ms   batched fwd+bwd %.3f ms" % (timeit(sep_fb), timeit(bat_fb)))
