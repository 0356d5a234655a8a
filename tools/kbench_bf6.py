"""dev tool: the bf16x6 GEMM (csrc/gemm_bf6.hip) on the step's 1x1-convolution shapes: error against an fp64 GEMM next to the error of the
library's fp32 convolution, and time against the library / the fp32-MFMA GEMM kernels.   python tools/kbench_bf6.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from distill_bev_amd import _lib as L
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
dev = torch.device("cuda:0")
def tm(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
TN = [0]            # tile width of the pack / launch pair (set per shape below)
def pack(w2):      # w2 [N, K]
    N, K = w2.shape
    nb = int(L.call("dbev_gemm_bf16x6_packed_bytes", N, K))
    p = torch.empty((nb,), dtype=torch.uint8, device=dev)
    L.call("dbev_gemm_bf16x6_pack", L.ptr(w2), w2.stride(0), w2.stride(1), N, K, TN[0], L.ptr(p), L.stream_ptr(dev))
    return p
def gemm(x, p, N):
    n, K, H, W = x.shape
    y = torch.empty((n, N, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    L.call("dbev_gemm_bf16x6_forward", L.ptr(x), L.ptr(p), L.ptr(y), n * H * W, K, N, K, TN[0], L.stream_ptr(dev))
    return y
SHAPES = [(48, 64, 256, 64, 176), (48, 256, 64, 64, 176), (48, 128, 512, 32, 88), (48, 512, 128, 32, 88), (48, 256, 1024, 16, 44),
          (48, 1024, 256, 16, 44), (48, 512, 2048, 8, 22), (48, 2048, 512, 8, 22), (48, 1024, 512, 16, 44), (8, 256, 256, 128, 128),
          (8, 512, 512, 64, 64), (48, 256, 128, 64, 176)]
if len(sys.argv) > 1: SHAPES = SHAPES[:int(sys.argv[1])]
print("shape (N, Cin, Cout, H, W)          GF | bf16x6 us  TF | miopen us  TF | gemm1x1 us | err bf16x6 / miopen (max |y - fp64| / max |y|)")
for (n, ci, co, h, w_) in SHAPES:
    g = torch.Generator().manual_seed(ci * 7 + co)
    x = torch.relu(torch.randn((n, ci, h, w_), generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((co, ci, 1, 1), generator=g) / ci ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    w2 = w.reshape(co, ci)
    from distill_bev_amd.gemm_bf6 import tile_n
    TN[0] = tile_n(n * h * w_, co)
    p = pack(w2)
    y = gemm(x, p, co)
    ym = F.conv2d(x, w)
    # fp64 reference on a slice of rows (the whole tensor in fp64 is slow for the big ones)
    xs = x.permute(0, 2, 3, 1).reshape(-1, ci)[:: max(1, (n * h * w_) // 8192)]
    ref = xs.double() @ w2.double().t()
    ys = y.permute(0, 2, 3, 1).reshape(-1, co)[:: max(1, (n * h * w_) // 8192)]
    yms = ym.permute(0, 2, 3, 1).reshape(-1, co)[:: max(1, (n * h * w_) // 8192)]
    sc = float(ref.abs().max())
    e1, e2 = float((ys.double() - ref).abs().max()) / sc, float((yms.double() - ref).abs().max()) / sc
    gf = 2.0 * n * h * w_ * ci * co / 1e9
    t1 = tm(lambda: gemm(x, p, co)); t2 = tm(lambda: F.conv2d(x, w))
    try:
        yg = torch.empty_like(y)
        t3 = tm(lambda: L.call("dbev_gemm1x1_forward", L.ptr(x), L.ptr(w2), L.ptr(yg), L.ptr(None), n * h * w_, ci, co, ci, L.stream_ptr(dev)))
    except Exception:
        t3 = float("nan")
    tp = tm(lambda: pack(w2))
    print(f"{str((n, ci, co, h, w_)):32s} {gf:6.1f} | {t1:8.1f} {gf / t1 * 1e3:6.1f} | {t2:8.1f} {gf / t2 * 1e3:6.1f} | {t3:8.1f} | {e1:.2e} / {e2:.2e}   pack {tp:.1f} us")

print("\nweight gradient: shape | bf16x6 us  TF | miopen us | err bf16x6 / miopen vs fp64")
from distill_bev_amd import gemm_bf6 as G
for (n, ci, co, h, w_) in SHAPES:
    g = torch.Generator().manual_seed(ci + co)
    x = torch.relu(torch.randn((n, ci, h, w_), generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((n, co, h, w_), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = torch.zeros((co, ci, 1, 1), device=dev).contiguous(memory_format=torch.channels_last)
    lib = lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    a = G.weight_gradient(x, gy, w)
    b = lib()
    step = max(1, (n * h * w_) // 65536)
    # fp64 reference of the full sum is affordable: [M, co]^T [M, ci]
    ref = gy.permute(0, 2, 3, 1).reshape(-1, co).double().t() @ x.permute(0, 2, 3, 1).reshape(-1, ci).double()
    sc = float(ref.abs().max())
    e1, e2 = float((a.reshape(co, ci).double() - ref).abs().max()) / sc, float((b.reshape(co, ci).double() - ref).abs().max()) / sc
    gf = 2.0 * n * h * w_ * ci * co / 1e9
    t1, t2 = tm(lambda: G.weight_gradient(x, gy, w)), tm(lib)
    print(f"{str((n, ci, co, h, w_)):32s} | {t1:8.1f} {gf / t1 * 1e3:6.1f} | {t2:8.1f} | {e1:.2e} / {e2:.2e}")
