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
Transposed convolutions (conv.py:300-346, indice.h:88-140 getValidOutPosTranspose: out = in * stride - padding + k * dilation,
out shape ops.py:33-44) the same two ways (`sparse_deconv_dense` through torch conv_transpose3d).  Max pooling
(pool_ops.h:26-98, maxpool_cuda.cu:28-230) straight from the pair lists: the output starts at ZERO and is raised by every paired
input; the backward gives every input equal to its output's value that output's gradient.
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


def deconv_out_shape(in_shape, ks, st, pd, dl, op):
    return [(in_shape[i] - 1) * st[i] - 2 * pd[i] + ks[i] + op[i] for i in range(3)]


def sparse_deconv_dense(features, indices, spatial_shape, batch_size, weight, bias=None, stride=1, padding=0, dilation=1,
                        out_padding=0):
    """SparseConvTranspose3d through a dense fp64 conv_transpose3d; active outputs = every cell some input reaches.  The reference's
    output extent is (in - 1) * stride - 2 * padding + ksize + out_padding (NO dilation term, ops.py:40-41): cells of the dense
    result beyond it are cut."""
    ks = list(weight.shape[:3])
    st, pd, dl, op = _t3(stride), _t3(padding), _t3(dilation), _t3(out_padding)
    f = torch.as_tensor(features, dtype=torch.float64)
    idx = torch.as_tensor(indices).long()
    Cin = weight.shape[3]
    D, H, W = [int(v) for v in spatial_shape]
    dense = torch.zeros((batch_size, Cin, D, H, W), dtype=torch.float64)
    occ = torch.zeros((batch_size, 1, D, H, W), dtype=torch.float64)
    dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = f
    occ[idx[:, 0], 0, idx[:, 1], idx[:, 2], idx[:, 3]] = 1.0
    w = torch.as_tensor(weight, dtype=torch.float64).permute(3, 4, 0, 1, 2).contiguous()        # [Cin, Cout, kz, ky, kx]
    y = F.conv_transpose3d(dense, w, None, st, pd, op, 1, dl)
    a = F.conv_transpose3d(occ, torch.ones((1, 1, *ks), dtype=torch.float64), None, st, pd, op, 1, dl)
    osh = deconv_out_shape([D, H, W], ks, st, pd, dl, op)
    full = torch.zeros((batch_size, y.shape[1], *osh), dtype=torch.float64)
    act = torch.zeros((batch_size, *osh), dtype=torch.bool)
    c = [min(osh[i], y.shape[2 + i]) for i in range(3)]
    full[:, :, :c[0], :c[1], :c[2]] = y[:, :, :c[0], :c[1], :c[2]]
    act[:, :c[0], :c[1], :c[2]] = a[:, 0, :c[0], :c[1], :c[2]] > 0
    oi = act.nonzero()
    out = full[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3]]
    if bias is not None:
        out = out + torch.as_tensor(bias, dtype=torch.float64)
    return out, oi, osh


def rulebook_pairs(indices, spatial_shape, batch_size, ksize, stride, padding, dilation, subm, transpose=False, out_padding=0):
    """-> (out_indices int64 [M, 4] ascending, pairs: list over the K offsets of int64 [n_k, 2] = (input row, output row))."""
    ks, st, pd, dl = _t3(ksize), _t3(stride), _t3(padding), _t3(dilation)
    if subm:
        st, pd = [1, 1, 1], [k // 2 for k in ks]
    idx = np.asarray(indices).astype(np.int64)
    ish = [int(v) for v in spatial_shape]
    if transpose:
        osh = deconv_out_shape(ish, ks, st, pd, dl, _t3(out_padding))
    else:
        osh = ish if subm else out_shape(ish, ks, st, pd, dl)
    offsets = [(a, b, c) for a in range(ks[0]) for b in range(ks[1]) for c in range(ks[2])]
    cand = {}
    per_k = [[] for _ in offsets]
    for i, (b, z, y, x) in enumerate(idx):
        for k, off in enumerate(offsets):
            o = []
            for a, v in enumerate((z, y, x)):
                if transpose:
                    val = v * st[a] - pd[a] + off[a] * dl[a]
                    if val < 0 or val >= osh[a]:
                        o = None
                        break
                    o.append(val)
                    continue
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


def maxpool_from_pairs(features, pairs, n_out):
    """indice_maxpool: zero-initialised output raised by every paired input, offsets in order (pool_ops.h:34-56)."""
    f = np.asarray(features, dtype=np.float64)
    out = np.zeros((n_out, f.shape[1]))
    for pr in pairs:
        if len(pr):
            np.maximum.at(out, pr[:, 1], f[pr[:, 0]])
    return out


def maxpool_backward_from_pairs(features, out, grad_out, pairs):
    """indice_maxpool_backward: din[i] += dout[o] for every pair with in[i] == out[o] (maxpool_cuda.cu:165-196), offsets in order."""
    f = np.asarray(features, dtype=np.float64)
    y, g = np.asarray(out, dtype=np.float64), np.asarray(grad_out, dtype=np.float64)
    din = np.zeros_like(f)
    for pr in pairs:
        if len(pr):
            np.add.at(din, pr[:, 0], np.where(f[pr[:, 0]] == y[pr[:, 1]], g[pr[:, 1]], 0.0))
    return din
