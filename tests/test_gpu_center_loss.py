"""Fused CenterHead loss (csrc/center_loss.hip, dbev_centerhead_loss_*) vs the reference's op-by-op sequence
(CenterHead.loss with fused_loss=False: clip_sigmoid, GaussianFocalLoss, gather, L1Loss), same targets, fp32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CH = dict(reg=2, height=1, dim=3, rot=2, vel=2)
KEYS = list(CH) + ["heatmap"]


def _setup(B, n_boxes, seed, dup=False):
    from distill_bev_amd import synthetic as syn
    from distill_bev_amd.center_head import LiDARBoxes
    from distill_bev_amd.train_step import build_model
    m, _ = build_model(allow_synthetic_teacher=True)
    head = m.pts_bbox_head.to("cuda:0")
    rng = np.random.default_rng(seed)
    boxes, labels = [], []
    for i in range(B):
        b, lab = syn.gt_boxes(n_boxes[i], rng)
        if dup and len(b) >= 3:                     # three objects of one task on the same BEV pixel
            b[1, :2] = b[0, :2] + 0.01; b[2, :2] = b[0, :2] - 0.01
            lab[1] = lab[0]; lab[2] = lab[0]
        boxes.append(LiDARBoxes(b)); labels.append(torch.from_numpy(lab))
    return head, boxes, labels


def _leaves(head, B, seed, channels_last):
    """-> list over tasks of {head name: leaf tensor}: fake head outputs"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for names in head.class_names:
        d = {}
        for k, c in list(CH.items()) + [("heatmap", len(names))]:
            t = torch.randn((B, c, 128, 128), generator=g)
            if k == "heatmap":
                t = t * 3.0 - 2.0                   # logits beyond the 1e-4 clip on both sides
                t[0, 0, :2, :2] = torch.tensor([[-12.0, 12.0], [9.3, -9.3]])
            t = t.to("cuda:0")
            if channels_last:
                t = t.contiguous(memory_format=torch.channels_last)
            d[k] = t.requires_grad_(True)
        out.append(d)
    return out


def _preds(leaves):
    # "* 1.0": the reference applies sigmoid_ in place to the head output, which must not be a leaf
    return tuple([{k: v * 1.0 for k, v in d.items()}] for d in leaves)


@pytest.mark.parametrize("channels_last,dup", [(True, False), (False, False), (True, True)])
def test_fused_loss_matches_op_sequence(channels_last, dup):
    from distill_bev_amd.center_head import CenterHead
    head, boxes, labels = _setup(3, (30, 0, 45), 4, dup)
    la_leaf, lb_leaf = _leaves(head, 3, 11, channels_last), _leaves(head, 3, 11, channels_last)
    pa, pb = _preds(la_leaf), _preds(lb_leaf)
    if channels_last:
        pa = tuple([{k: v.contiguous(memory_format=torch.channels_last) for k, v in p[0].items()}] for p in pa)
    la = head.loss(boxes, labels, pa)
    try:
        CenterHead.fused_loss = False
        lb = head.loss(boxes, labels, pb)
    finally:
        CenterHead.fused_loss = True
    assert set(la) == set(lb) and len(la) == 36
    for k in lb:
        a, b = float(la[k].detach()), float(lb[k].detach())
        assert abs(a - b) <= 2e-6 * max(abs(b), 1e-3), (k, a, b)
    for t in range(6):                              # side effect of the reference kept: dict holds the clipped sigmoid
        assert torch.allclose(pa[t][0]["heatmap"], pb[t][0]["heatmap"], atol=1e-7)
        lo, hi = float(np.float32(1e-4)), float(np.float32(1) - np.float32(1e-4))
        assert float(pa[t][0]["heatmap"].min()) >= lo and float(pa[t][0]["heatmap"].max()) <= hi
    wts = {k: 0.5 + 0.1 * i for i, k in enumerate(sorted(la))}
    sum(wts[k] * la[k] for k in la).backward()
    sum(wts[k] * lb[k] for k in lb).backward()
    for t in range(6):
        for k in KEYS:
            ga, gb = la_leaf[t][k].grad, lb_leaf[t][k].grad
            assert ga is not None and gb is not None, (t, k)
            scale = float(gb.abs().max()) + 1e-12
            assert float((ga - gb).abs().max()) <= 2e-5 * scale, (t, k, float((ga - gb).abs().max()), scale)


def test_fused_loss_is_bit_reproducible():
    head, boxes, labels = _setup(2, (30, 30), 8, True)
    runs = []
    for _ in range(2):
        leaf = _leaves(head, 2, 5, True)
        l = head.loss(boxes, labels, _preds(leaf))
        sum(l.values()).backward()
        runs.append((l, leaf))
    for k in runs[0][0]:
        assert float(runs[0][0][k]) == float(runs[1][0][k])
    for t in range(6):
        for k in KEYS:
            assert torch.equal(runs[0][1][t][k].grad, runs[1][1][t][k].grad)


def test_batch_without_any_object():
    """no GT box in the whole batch: targets all zero, avg factors fall back to 1 / 1e-4, losses finite, gradients flow
    through the heat-map term only -- same as the op sequence."""
    from distill_bev_amd.center_head import CenterHead
    head, boxes, labels = _setup(2, (0, 0), 3)
    la_leaf, lb_leaf = _leaves(head, 2, 2, True), _leaves(head, 2, 2, True)
    la = head.loss(boxes, labels, _preds(la_leaf))
    try:
        CenterHead.fused_loss = False
        lb = head.loss(boxes, labels, _preds(lb_leaf))
    finally:
        CenterHead.fused_loss = True
    for k in lb:
        a, b = float(la[k].detach()), float(lb[k].detach())
        assert np.isfinite(a) and abs(a - b) <= 2e-6 * max(abs(b), 1e-3), (k, a, b)
        if "heatmap" not in k:
            assert a == 0.0
    sum(la.values()).backward(); sum(lb.values()).backward()
    for t in range(6):
        assert float(la_leaf[t]["reg"].grad.abs().max()) == 0.0
        ga, gb = la_leaf[t]["heatmap"].grad, lb_leaf[t]["heatmap"].grad
        assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max())
