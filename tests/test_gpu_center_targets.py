"""dbev_centerhead_targets (csrc/center_targets.hip) through CenterHead.get_targets vs the host restatement of the
reference's target assignment (oracle/center_targets.py, pinned against the imported gaussian.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _head(**over):
    from distill_bev_amd.train_step import build_model
    m, _ = build_model(allow_synthetic_teacher=True)
    h = m.pts_bbox_head
    h.train_cfg = dict(h.train_cfg); h.train_cfg.update(over)
    return h


def _ulps(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def _compare(head, boxes, labels):
    from oracle import center_targets as OCT
    dev = torch.device("cuda:0")
    ref = OCT.get_targets(head, boxes, labels, torch.device("cpu"))
    out = head.get_targets(boxes, labels, dev)
    for t in range(len(head.task_heads)):
        hm_r, hm_o = ref[0][t].numpy(), out[0][t].cpu().numpy()
        assert hm_o.shape == hm_r.shape
        # fp64 exp rounded to fp32: identical up to a last-bit difference of the two libm's
        assert _ulps(hm_o, hm_r).max() <= 1, t
        assert np.array_equal(out[2][t].cpu().numpy(), ref[2][t].numpy()), t            # ind (int64), exact
        assert np.array_equal(out[3][t].cpu().numpy(), ref[3][t].numpy()), t            # mask (uint8), exact
        ab_r, ab_o = ref[1][t].numpy(), out[1][t].cpu().numpy()
        assert np.array_equal(ab_o[..., [0, 1, 2, 8, 9]], ab_r[..., [0, 1, 2, 8, 9]]), t   # pure fp32 arithmetic: exact
        assert _ulps(ab_o[..., 3:8], ab_r[..., 3:8]).max() <= 2, t                      # logf / sinf / cosf
    return ref, out


def test_targets_match_host_restatement_on_synthetic_batches():
    from distill_bev_amd import synthetic as syn
    from distill_bev_amd.center_head import LiDARBoxes
    head = _head()
    rng = np.random.default_rng(3)
    boxes, labels = [], []
    for n in (30, 1, 0, 57):                        # an empty sample in the middle of the batch
        b, lab = syn.gt_boxes(n, rng)
        boxes.append(LiDARBoxes(b)); labels.append(torch.from_numpy(lab))
    ref, out = _compare(head, boxes, labels)
    assert sum(int(m.sum()) for m in out[3]) > 60
    assert float(out[0][1].max()) == 1.0


def test_targets_edge_cases_overflow_border_invalid():
    from distill_bev_amd.center_head import LiDARBoxes
    head = _head(max_objs=4)                        # more boxes of one task than slots -> the tail draws nothing
    rng = np.random.default_rng(5)
    n = 24
    b = np.zeros((n, 9), np.float32)
    b[:, 0] = rng.uniform(-60, 60, n); b[:, 1] = rng.uniform(-60, 60, n)     # some centres outside the 102.4 m range
    b[:, 2] = rng.uniform(-3, 1, n)
    b[:, 3:6] = rng.uniform(0.3, 12, (n, 3))
    b[:, 6] = rng.uniform(-3.2, 3.2, n); b[:, 7:9] = rng.normal(size=(n, 2))
    b[0, 3] = 0.0                                   # zero width: skipped
    b[1, 0], b[1, 1] = -51.25, 51.19                # trunc(-0.03) = -0 -> column 0 is still inside; top row border
    b[2, 0], b[2, 1] = 51.199, -51.2                # last column, first row
    lab = rng.integers(0, 10, n).astype(np.int64)
    lab[3] = -1                                     # ignored label
    lab[4:14] = 0                                   # 10 cars for 4 slots
    _compare(head, [LiDARBoxes(b)], [torch.from_numpy(lab)])


def test_targets_are_deterministic_and_reject_cpu():
    from distill_bev_amd import synthetic as syn
    from distill_bev_amd._lib import DbevHipError
    from distill_bev_amd.center_head import LiDARBoxes
    head = _head()
    b, lab = syn.gt_boxes(40, np.random.default_rng(9))
    args = ([LiDARBoxes(b)], [torch.from_numpy(lab)])
    a = head.get_targets(*args, torch.device("cuda:0"))
    c = head.get_targets(*args, torch.device("cuda:0"))
    for x, y in zip(a, c):
        for u, v in zip(x, y):
            assert torch.equal(u, v)
    with pytest.raises(DbevHipError):
        head.get_targets(*args, torch.device("cpu"))


def test_targets_match_fixture_drawn_by_the_reference_gaussian_utils():
    """tests/golden/center_targets.npz (make_golden.py 'center': imported core/utils/gaussian.py + the reference's loop)."""
    from conftest import load_golden
    from distill_bev_amd.center_head import LiDARBoxes
    fx = load_golden("center_targets.npz")
    head = _head()
    boxes, labels = [], []
    for b in range(2):
        g9 = fx[f"boxes{b}"].copy()
        bottom = g9.copy(); bottom[:, 2] = g9[:, 2] - g9[:, 5] * 0.5
        boxes.append(LiDARBoxes(bottom)); labels.append(torch.from_numpy(fx[f"labels{b}"]))
    hms, abox, inds, masks = head.get_targets(boxes, labels, torch.device("cuda:0"))
    edges = np.cumsum([0] + [len(n) for n in head.class_names])
    for t in range(6):
        assert np.array_equal(masks[t].cpu().numpy(), fx["mask"][t])
        assert np.array_equal(inds[t].cpu().numpy(), fx["ind"][t])
        assert _ulps(hms[t].cpu().numpy(), fx["heatmap"][:, edges[t]:edges[t + 1]]).max() <= 1
        u = _ulps(abox[t].cpu().numpy(), fx["anno_box"][t])
        assert u[..., [0, 1, 8, 9]].max() == 0 and u.max() <= 2
