"""CPU: the two restatements of sparse convolution in oracle/spconv.py (dense conv3d definition vs the rulebook pair lists)
agree on submanifold, strided, asymmetric-kernel, dilated and transposed cases; the pair-list max pooling against a brute-force
window maximum."""
import numpy as np
import pytest
import torch

from oracle import spconv as OS


def _cloud(seed, B, shape, n, C):
    rng = np.random.default_rng(seed)
    vol = B * shape[0] * shape[1] * shape[2]
    lin = rng.choice(vol, size=n, replace=False)
    rng.shuffle(lin)                                           # arbitrary row order
    b, r = np.divmod(lin, shape[0] * shape[1] * shape[2])
    z, r = np.divmod(r, shape[1] * shape[2])
    y, x = np.divmod(r, shape[2])
    return np.stack([b, z, y, x], 1).astype(np.int32), rng.normal(size=(n, C)).astype(np.float32)


@pytest.mark.parametrize("ks,st,pd,dl,subm", [((3, 3, 3), 1, 1, 1, True), ((3, 3, 3), 2, 1, 1, False),
                                              ((3, 1, 1), (2, 1, 1), 0, 1, False), ((3, 3, 3), 2, (0, 1, 1), 1, False),
                                              ((3, 3, 3), 1, 0, 1, True), ((3, 3, 3), 1, 2, 2, False)])
def test_dense_definition_equals_rulebook(ks, st, pd, dl, subm):
    idx, feats = _cloud(3, 2, (5, 8, 7), 90, 4)
    w = np.random.default_rng(4).normal(size=(*ks, 4, 6))
    ref, oi, osh = OS.sparse_conv_dense(feats, idx, (5, 8, 7), 2, w, None, st, pd, dl, subm)
    out_idx, pairs = OS.rulebook_pairs(idx, (5, 8, 7), 2, ks, st, pd, dl, subm)
    got = OS.conv_from_pairs(feats, w.reshape(-1, 4, 6), pairs, out_idx.shape[0])
    if subm:                                                   # rows follow the input order; the dense path sorts
        order = np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1], idx[:, 0]))
        assert np.array_equal(idx[order].astype(np.int64), oi.numpy())
        got = got[order]
    else:
        assert np.array_equal(out_idx, oi.numpy())
    assert np.abs(got - ref.numpy()).max() < 1e-12
    assert sum(len(p) for p in pairs) > 90


@pytest.mark.parametrize("ks,st,pd,dl,op", [((3, 3, 3), 2, 1, 1, 0), ((3, 3, 3), 2, 1, 1, 1), ((2, 2, 2), 2, 0, 1, 0),
                                            ((3, 1, 3), (1, 2, 2), (1, 0, 1), 1, 0), ((3, 3, 3), 1, 0, 2, 0)])
def test_transposed_dense_definition_equals_rulebook(ks, st, pd, dl, op):
    idx, feats = _cloud(5, 2, (4, 6, 5), 50, 3)
    w = np.random.default_rng(6).normal(size=(*ks, 3, 5))
    ref, oi, osh = OS.sparse_deconv_dense(feats, idx, (4, 6, 5), 2, w, None, st, pd, dl, op)
    out_idx, pairs = OS.rulebook_pairs(idx, (4, 6, 5), 2, ks, st, pd, dl, False, True, op)
    assert np.array_equal(out_idx, oi.numpy()) and len(out_idx) > 50
    got = OS.conv_from_pairs(feats, w.reshape(-1, 3, 5), pairs, out_idx.shape[0])
    assert np.abs(got - ref.numpy()).max() < 1e-12
    for pr in pairs:                                           # an (input, offset) pair reaches exactly one output
        assert len(np.unique(pr[:, 0])) == len(pr)


def test_maxpool_pairs_equal_brute_force_window_maximum_and_tie_gradients():
    shape, B = (4, 6, 6), 2
    idx, feats = _cloud(9, B, shape, 70, 4)
    feats = np.round(feats * 2) / 2                            # ties
    out_idx, pairs = OS.rulebook_pairs(idx, shape, B, 3, 2, 1, 1, False)
    got = OS.maxpool_from_pairs(feats, pairs, len(out_idx))
    cell = {tuple(r): i for i, r in enumerate(idx.tolist())}
    ref = np.zeros_like(got)
    for o, (b, z, y, x) in enumerate(out_idx.tolist()):
        for kz in range(3):
            for ky in range(3):
                for kx in range(3):
                    i = cell.get((b, 2 * z - 1 + kz, 2 * y - 1 + ky, 2 * x - 1 + kx))
                    if i is not None:
                        ref[o] = np.maximum(ref[o], feats[i])
    assert np.array_equal(got, ref) and (got == 0).any() and (got > 0).any()
    g = np.random.default_rng(1).normal(size=got.shape)
    din = OS.maxpool_backward_from_pairs(feats, got, g, pairs)
    # every output's gradient reaches each of its inputs that equals the output value (zero outputs reach inputs that are exactly 0)
    tot = 0.0
    for pr in pairs:
        tot += np.where(feats[pr[:, 0]] == got[pr[:, 1]], g[pr[:, 1]], 0.0).sum()
    assert abs(din.sum() - tot) < 1e-9 and np.abs(din).max() > 0
