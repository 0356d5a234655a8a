"""CPU: oracle/distill.py geometry half against the fixtures produced by the imported
reference box_np_ops.points_in_rbbox (driven as bevdet_distill.py:755-843 drives it)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import distill as OD


@pytest.mark.parametrize("H", [128, 64])
def test_foreground_scale_mask_bit_exact(H):
    g = load_golden(f"fgmask_{H}.npz")
    boxes = [g["boxes0"], g["boxes1"], g["boxes2"]]
    assert np.array_equal(OD.cell_coords(H, 0.1, 1024 // H, -51.2), g["xs"])
    fg, fs, bs = OD.foreground_scale_mask(H, H, boxes)
    assert np.array_equal(fg, g["fg"])
    # reference fg_scale = torch.sqrt (CPU kernel, not correctly rounded, host dependent): 1 ulp
    assert np.allclose(fs, g["fg_scale"], rtol=2.5e-7, atol=0)
    assert np.array_equal(bs, g["bg_scale"])


def test_points_in_rbbox_mask_bit_exact():
    g = load_golden("fgmask_128.npz")
    for k in (0, 1):
        b = g[f"boxes{k}"][:, :7].copy()
        b[:, 2] = 0; b[:, 5] = 1
        m = OD.points_in_rbbox(g["points"], b)
        ref = np.unpackbits(g[f"mask{k}"])[: m.size].reshape(m.shape).astype(bool)
        assert np.array_equal(m, ref)


def test_fgd_losses_brute_force_tiny():
    """Loss arithmetic (parity unpinned by reference outputs): hand-computed tiny case."""
    rng = np.random.default_rng(0)
    B, C, H, W = 2, 3, 4, 4
    S = rng.normal(size=(B, C, H, W)).astype(np.float32)
    T = rng.normal(size=(B, C, H, W)).astype(np.float32)
    fg = np.zeros((B, 1, H, W), np.float32); fg[0, 0, 1, 1] = 1; fg[1, 0, 2, :2] = 1
    fs = fg * 0.5
    bs = np.stack([np.full((1, H, W), 1.0 / (H * W - fg[b].sum()), np.float32) for b in range(B)])
    out, aux = OD.fgd_feature_losses(S, T, fg, fs, bs)
    # brute force
    tot_fg = tot_bg = 0.0
    for b in range(B):
        ta = np.abs(T[b].astype(np.float64)).mean(0).reshape(-1) / 0.5
        sa = np.abs(S[b].astype(np.float64)).mean(0).reshape(-1) / 0.5
        ta = np.exp(ta) / np.exp(ta).sum() * H * W
        sa = np.exp(sa) / np.exp(sa).sum() * H * W
        att = ((ta + sa) / 2).reshape(H, W)
        for y in range(H):
            for x in range(W):
                sc = max(float(fs[b, 0, y, x]), float(bs[b, 0, y, x]))
                for c in range(C):
                    sq = (float(S[b, c, y, x]) - float(T[b, c, y, x])) ** 2
                    if fg[b, 0, y, x]:
                        tot_fg += sq * sc * att[y, x]
                    else:
                        tot_bg += sq * sc * att[y, x]
    assert abs(out["kd_fg_feat_loss"] - tot_fg * 6e-3 / B) < 1e-12
    assert abs(out["kd_bg_feat_loss"] - tot_bg * 4e-2 / B) < 1e-12


def test_center_targets_restatement_matches_fixture_from_imported_gaussian_utils():
    """oracle/center_targets.py vs tests/golden/center_targets.npz (heat maps drawn by the reference's own
    core/utils/gaussian.py, slot/regression logic of centerpoint_head.py:447-611; make_golden.py 'center')."""
    import torch
    from conftest import load_golden
    from distill_bev_amd.center_head import LiDARBoxes
    from distill_bev_amd.train_step import build_model
    from oracle import center_targets as OCT
    fx = load_golden("center_targets.npz")
    head = build_model(allow_synthetic_teacher=True)[0].pts_bbox_head
    boxes, labels = [], []
    for b in range(2):
        g9 = fx[f"boxes{b}"].copy()
        bottom = g9.copy(); bottom[:, 2] = g9[:, 2] - g9[:, 5] * 0.5      # LiDARBoxes holds bottom centres
        boxes.append(LiDARBoxes(bottom)); labels.append(torch.from_numpy(fx[f"labels{b}"]))
    hms, abox, inds, masks = OCT.get_targets(head, boxes, labels, torch.device("cpu"))
    edges = np.cumsum([0] + [len(n) for n in head.class_names])
    for t in range(6):
        assert np.array_equal(masks[t].numpy(), fx["mask"][t]) and np.array_equal(inds[t].numpy(), fx["ind"][t])
        hm = hms[t].numpy(); ref = fx["heatmap"][:, edges[t]:edges[t + 1]]
        assert np.array_equal(hm, ref)
        a, r = abox[t].numpy(), fx["anno_box"][t]
        ulp = np.abs(a.view(np.int32).astype(np.int64) - r.view(np.int32).astype(np.int64))
        # z re-derived from bottom + h/2 and torch-vs-numpy float32 log/sin/cos: last-bit differences only
        assert ulp[..., [0, 1, 8, 9]].max() == 0 and ulp.max() <= 2
