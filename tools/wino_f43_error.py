"""Would Winograd F(4x4, 3x3) be accurate enough for the step's 3x3 layers?  (cost it before any kernel.)

fp32 emulation on the CPU of the three algorithms' arithmetic on the fifteen layer shapes of tools/kbench_wino.py (batch and map
cropped: the error depends on the reduction length C and the data, not on how many tiles there are):
  * F(2x2, 3x3)   -- what csrc/wino.hip computes (points 0, +-1, inf);
  * F(4x4, 3x3)   -- Lavin & Gray's points 0, +-1, +-2, inf;
  * F(4x4, 3x3)h  -- points 0, +-1, +-1/2, inf (smaller transform constants);
  * direct fp32   -- torch's CPU convolution, standing in for the library's direct kernels.
Filter transform in fp64 rounded once to fp32 (an offline pack kernel can afford that), input / output transforms and the channel
contraction in fp32.  Reported: max |y - y64| / max |y64| against an fp64 convolution, per layer, and the ratio to F(2x2).
python tools/wino_f43_error.py > profiles/r06_wino_f43_error.txt
"""
import itertools
import sys
import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(0)


def cook_toom(points, m, r):
    """A^T [m, n], G [n, r], B^T [n, n] (float64) for F(m, r) on the finite `points` + infinity, n = m + r - 1"""
    n = m + r - 1
    a = np.array(points, dtype=np.float64)
    assert len(a) == n - 1
    At = np.zeros((m, n)); G = np.zeros((n, r))
    for j in range(n - 1):
        f = np.prod([a[j] - a[k] for k in range(n - 1) if k != j])
        At[:, j] = a[j] ** np.arange(m)
        G[j] = a[j] ** np.arange(r) / f
    At[m - 1, n - 1] = 1.0
    G[n - 1, r - 1] = 1.0
    # B^T from the identity  y = A^T [(G g) * (B^T d)]  for all g, d  (y_i = sum_l d_{i + l} g_l)
    Bt = np.zeros((n, n))
    for k in range(n):
        rows, rhs = [], []
        for l in range(r):
            rows.append(At * G[:, l][None, :])
            rhs.append(np.array([1.0 if k == i + l else 0.0 for i in range(m)]))
        sol, *_ = np.linalg.lstsq(np.concatenate(rows), np.concatenate(rhs), rcond=None)
        Bt[:, k] = sol
    Bt = np.round(Bt * 64) / 64 if np.abs(np.round(Bt * 64) / 64 - Bt).max() < 1e-9 else Bt
    return At, G, Bt


def check(At, G, Bt, m, r):
    d, g = np.random.rand(m + r - 1), np.random.rand(r)
    y = At @ ((G @ g) * (Bt @ d))
    ref = np.array([sum(d[i + l] * g[l] for l in range(r)) for i in range(m)])
    assert np.abs(y - ref).max() < 1e-12, np.abs(y - ref).max()


def wino_conv(x, w, At, G, Bt, m):
    """x [N, C, H, W] fp32, w [Co, C, 3, 3] fp32 -> y fp32 through F(m x m, 3 x 3) with fp32 transforms and contraction"""
    N, C, H, W = x.shape
    n = m + 2
    Hp, Wp = -(-H // m) * m, -(-W // m) * m
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    t = xp.unfold(2, n, m).unfold(3, n, m)                               # [N, C, th, tw, n, n]
    Btf, Atf = torch.from_numpy(Bt).float(), torch.from_numpy(At).float()
    U = torch.einsum("ia,ocab,jb->ijoc", torch.from_numpy(G), w.double(), torch.from_numpy(G)).float()      # fp64, rounded once
    V = torch.einsum("ia,nchwab->nchwib", Btf, t)                        # rows, fp32
    V = torch.einsum("nchwib,jb->ijnhwc", V, Btf).contiguous()           # columns, fp32
    th, tw = V.shape[3], V.shape[4]
    M = torch.matmul(V.reshape(n, n, N * th * tw, C), U.transpose(2, 3))  # [n, n, tiles, Co] fp32 contraction
    Y = torch.einsum("ia,abto->ibto", Atf, M)
    Y = torch.einsum("ibto,jb->tijo", Y, Atf)                            # [tiles, m, m, Co]
    Co = w.shape[0]
    Y = Y.reshape(N, th, tw, m, m, Co).permute(0, 5, 1, 3, 2, 4).reshape(N, Co, th * m, tw * m)
    return Y[:, :, :H, :W]


SHAPES = [(48, 256, 256, 16, 44), (48, 128, 128, 32, 88), (48, 64, 64, 64, 176), (48, 512, 512, 8, 22), (8, 64, 64, 128, 128),
          (8, 512, 256, 128, 128), (8, 128, 128, 128, 128), (8, 256, 256, 64, 64), (8, 512, 512, 64, 64), (8, 640, 512, 64, 64),
          (8, 64, 2304, 128, 128), (8, 64, 64, 256, 256), (8, 256, 256, 32, 32), (8, 512, 512, 16, 16), (48, 512, 512, 16, 44)]

algos = {"F(2,3)": (cook_toom([0, 1, -1], 2, 3), 2), "F(4,3)": (cook_toom([0, 1, -1, 2, -2], 4, 3), 4),
         "F(4,3)h": (cook_toom([0, 1, -1, 0.5, -0.5], 4, 3), 4)}
for (At, G, Bt), m in algos.values():
    check(At, G, Bt, m, 3)
print(__doc__.split("python tools")[0].strip().replace("\n", "\n# ").join(["# ", ""]))
print("# inputs: x = relu(randn) (what a 3x3 layer sees behind norm + ReLU) and x = randn (a gradient map, the data-gradient use), w = randn / (3 sqrt C)")
print("%-28s %-6s | %-9s %-9s %-9s %-9s | %-8s %-8s" % ("shape (N, C, Co, H, W)", "input", "direct", "F(2,3)", "F(4,3)", "F(4,3)h", "F43/F23", "F43h/F23"))
worst = {k: 0.0 for k in algos}
for (N, C, Co, H, W) in SHAPES:
    n_, co_, h_, w_ = min(N, 2), min(Co, 128), min(H, 32), min(W, 44)
    for kind in ("relu", "randn"):
        x = torch.randn((n_, C, h_, w_))
        if kind == "relu":
            x = torch.relu(x)
        w = torch.randn((co_, C, 3, 3)) / (3 * C ** 0.5)
        y64 = F.conv2d(x.double(), w.double(), padding=1)
        scale = float(y64.abs().max())
        err = {"direct": float((F.conv2d(x, w, padding=1).double() - y64).abs().max()) / scale}
        for name, ((At, G, Bt), m) in algos.items():
            err[name] = float((wino_conv(x, w, At, G, Bt, m).double() - y64).abs().max()) / scale
            worst[name] = max(worst[name], err[name] / err["F(2,3)"]) if name != "F(2,3)" else 0.0
        print("%-28s %-6s | %-9.2e %-9.2e %-9.2e %-9.2e | %-8.1f %-8.1f" % ((N, C, Co, H, W), kind, err["direct"], err["F(2,3)"], err["F(4,3)"],
              err["F(4,3)h"], err["F(4,3)"] / err["F(2,3)"], err["F(4,3)h"] / err["F(2,3)"]))
        sys.stdout.flush()
print("# worst ratio to F(2,3): F(4,3) %.1f x, F(4,3)h %.1f x" % (worst["F(4,3)"], worst["F(4,3)h"]))
