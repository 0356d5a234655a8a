"""CPU: the two restatements of sparse convolution in oracle/spconv.py (dense conv3d definition vs the rulebook pair lists)
agree on submanifold, strided, asymmetric-kernel and dilated cases."""
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
