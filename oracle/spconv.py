"""Oracle (CPU, torch fp64 / numpy) for the sparse convolutions of the voxel teachers -- TEST INFRASTRUCTURE ONLY.

The reference's implementation is its bundled spconv v1.x CUDA extension (mmdet3d/ops/spconv/src/*.cu,
include/spconv/*.h); src/all.cc:15 hard-includes cuda_runtime_api.h, so it cannot be built in this container (no CUDA):
PARITY UNPINNED by reference outputs.  Restated from the source two independent ways that tests/test_oracle_spconv.py
checks against each other:
  * `sparse_conv_dense`   definition through a DENSE torch conv3d in fp64: out = conv3d(densified input) evaluated at the
                          active output sites; active sites = the input sites for a submanifold convolution
                          (spconv_ops.h:76-80: stride 1, padding ksize // 2 whatever the layer was configured with), else
                          every output cell whose receptive field contains an active input (indice.cu.h:24-66); sites
                          in ascending (batch, z, y, x) order (spconv_ops.h: torch::_unique of the output cell ids);
  * `rulebook_pairs`      the (input row, output row) pair lists per kernel offset straight from the rule
                          in = out * stride - padding + k * dilation  (indice.h getValidOutPos), offsets in (kz, ky, kx)
                          row-major order = the weight's [kz, ky, kx, Cin, Cout] layout (conv.py:108-109).
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t3(v):
    return [int(x) for x in v] if isinstance(v, (list, tuple)) else [int(v)] * 3


def out_shape(in_shape, ks, st, pd, dl):
    return [(in_shape[i] + 2 * pd[i] - dl[i] * (ks[i] - 1) - 1) // st[i] + 1 for i in range(3)]


def sparse_conv_dense(features, indices, spatial_shape, batch_size, weight, bias=None, stride=1, padding=0, dilation=1,
                      subm=False):
    """features [N, Cin], indices int [N, 4] (b, z, y, x), weight [kz, ky, kx, Cin, Cout]
    -> (out_features fp64 [M, Cout], out_indices int64 [M, 4], out_spatial_shape)."""
    ks = list(weight.shape[:3])
    st, pd, dl = _t3(stride), _t3(padding), _t3(dilation)
    if subm:
        st, pd = [1, 1, 1], [k // 2 for k in ks]
    f = torch.as_tensor(features, dtype=torch.float64)
    idx = torch.as_tensor(indices).long()
    Cin, Cout = weight.shape[3], weight.shape[4]
    D, H, W = [int(v) for v in spatial_shape]
    dense = torch.zeros((batch_size, Cin, D, H, W), dtype=torch.float64)
    occ = torch.zeros((batch_size, 1, D, H, W), dtype=torch.float64)
    dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = f
    occ[idx[:, 0], 0, idx[:, 1], idx[:, 2], idx[:, 3]] = 1.0
    w = torch.as_tensor(weight, dtype=torch.float64).permute(4, 3, 0, 1, 2).contiguous()
    y = F.conv3d(dense, w, None, st, pd, dl)
    if subm:
        act = occ[:, 0] > 0
        oshape = [D, H, W]
    else:
        act = F.conv3d(occ, torch.ones((1, 1, *ks), dtype=torch.float64), None, st, pd, dl)[:, 0] > 0
        oshape = list(y.shape[2:])
    oi = act.nonzero()                                   # ascending (b, z, y, x)
    out = y[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3]]
    if bias is not None:
        out = out + torch.as_tensor(bias, dtype=torch.float64)
    return out, oi, oshape


def rulebook_pairs(indices, spatial_shape, batch_size, ksize, stride, padding, dilation, subm):
    """-> (out_indices int64 [M, 4] ascending, pairs: list over the K offsets of int64 [n_k, 2] = (input row, output row))."""
    ks, st, pd, dl = _t3(ksize), _t3(stride), _t3(padding), _t3(dilation)
    if subm:
        st, pd = [1, 1, 1], [k // 2 for k in ks]
    idx = np.asarray(indices).astype(np.int64)
    ish = [int(v) for v in spatial_shape]
    osh = ish if subm else out_shape(ish, ks, st, pd, dl)
    offsets = [(a, b, c) for a in range(ks[0]) for b in range(ks[1]) for c in range(ks[2])]
    cand = {}
    per_k = [[] for _ in offsets]
    for i, (b, z, y, x) in enumerate(idx):
        for k, off in enumerate(offsets):
            o = []
            for a, v in enumerate((z, y, x)):
                num = v + pd[a] - off[a] * dl[a]
                if num < 0 or num % st[a] or num // st[a] >= osh[a]:
                    o = None
                    break
                o.append(num // st[a])
            if o is None:
                continue
            key = (int(b), *o)
            cand.setdefault(key, None)
            per_k[k].append((i, key))
    if subm:
        keys = [tuple(int(v) for v in r) for r in idx]
        order = {k: n for n, k in enumerate(keys)}
        out_idx = idx
        per_k = [[(i, key) for (i, key) in lst if key in order] for lst in per_k]
    else:
        keys = sorted(cand)
        order = {k: n for n, k in enumerate(keys)}
        out_idx = np.array(keys, dtype=np.int64).reshape(-1, 4)
    pairs = [np.array([(i, order[key]) for (i, key) in lst], dtype=np.int64).reshape(-1, 2) for lst in per_k]
    return out_idx, pairs


def conv_from_pairs(features, weight, pairs, n_out, bias=None):
    """out[o] = bias + sum_k sum_{(i, o) in pairs[k]} features[i] @ weight[k]   (fp64; weight [K, Cin, Cout])."""
    f = np.asarray(features, dtype=np.float64)
    w = np.asarray(weight, dtype=np.float64)
    out = np.zeros((n_out, w.shape[-1]))
    for k, pr in enumerate(pairs):
        if len(pr):
            np.add.at(out, pr[:, 1], f[pr[:, 0]] @ w[k])
    return out if bias is None else out + np.asarray(bias, dtype=np.float64)
