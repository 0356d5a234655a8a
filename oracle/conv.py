"""TEST INFRASTRUCTURE -- CPU restatement of the dense convolution the reference's blocks run (`nn.Conv2d(k=3, s=1, p=1)` and
`nn.Conv2d(k=1)` -> cuDNN: mmdet3d/models/bricks/res_block.py:11-230, necks/lss_fpn.py:30-60, dense_heads/centerpoint_head.py:17-130,
backbones/second.py:60-78), written out from its definition (a sum over taps and input channels, fp64 numpy), and of the Winograd
F(2x2, 3x3) identity the HIP kernels (distill_bev_amd/csrc/wino.hip) evaluate.  Pinned: `conv2d_direct` equals
torch.nn.functional.conv2d (the op the reference calls) in tests/test_oracle_conv.py; `conv3x3_winograd` equals `conv2d_direct`.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def conv2d_direct(x, w, bias=None, padding=None):
    """y[n, o, i, j] = b[o] + sum_{c, a, b} x[n, c, i + a - p, j + b - p] * w[o, c, a, b]  (stride 1, zero padding p = k // 2)"""
    x, w = np.asarray(x, np.float64), np.asarray(w, np.float64)
    N, C, H, W = x.shape
    O, _, kh, kw = w.shape
    p = kh // 2 if padding is None else padding
    xp = np.zeros((N, C, H + 2 * p, W + 2 * p))
    xp[:, :, p:p + H, p:p + W] = x
    Ho, Wo = H + 2 * p - kh + 1, W + 2 * p - kw + 1
    y = np.zeros((N, O, Ho, Wo))
    for a in range(kh):
        for b in range(kw):
            y += np.einsum("nchw,oc->nohw", xp[:, :, a:a + Ho, b:b + Wo], w[:, :, a, b])
    if bias is not None:
        y += np.asarray(bias, np.float64)[None, :, None, None]
    return y


def conv3x3_winograd(x, w):
    """the same 3x3 / stride-1 / pad-1 convolution through Y = A^T [ (G g G^T) .* (B^T d B) ] A per 2x2 output tile (H, W even)"""
    x, w = np.asarray(x, np.float64), np.asarray(w, np.float64)
    N, C, H, W = x.shape
    O = w.shape[0]
    xp = np.zeros((N, C, H + 2, W + 2))
    xp[:, :, 1:-1, 1:-1] = x
    U = np.einsum("ia,ocab,jb->ocij", G, w, G)                      # [O, C, 4, 4]
    y = np.zeros((N, O, H, W))
    for th in range(H // 2):
        for tw in range(W // 2):
            d = xp[:, :, 2 * th:2 * th + 4, 2 * tw:2 * tw + 4]      # [N, C, 4, 4]
            V = np.einsum("ia,ncab,jb->ncij", BT, d, BT)
            M = np.einsum("ncij,ocij->noij", V, U)
            y[:, :, 2 * th:2 * th + 2, 2 * tw:2 * tw + 2] = np.einsum("ai,noij,bj->noab", AT, M, AT)
    return y
