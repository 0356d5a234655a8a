"""CPU: the host-side pieces of the BEVFormer path that need no kernel -- against the same reference-generated fixtures the GPU
tests use (tests/golden/make_golden.py bevformer): set-prediction targets + losses (Hungarian matching, focal / L1) and box
decoding from the fixture's own head outputs, the camera projection of the pillar points, GridMask's numpy draws; plus
properties of the torchvision-free rotation (the one brick of this path no reference output pins)."""
import os
import sys

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)


def _head(name="bevformer_head.npz", cfg="small_bevformer_head_cfg"):
    import bevformer_cfgs as C
    from distill_bev_amd import bevformer  # noqa: F401
    from distill_bev_amd.registry import build_head
    fx = np.load(os.path.join(GOLD, name))
    head = build_head(getattr(C, cfg)())
    head.load_state_dict({k[6:].replace("__", "."): torch.from_numpy(fx[k]) for k in fx.files if k.startswith("head__")}, strict=True)
    return head.eval(), fx


def _gts(fx, bs):
    from distill_bev_amd.center_head import LiDARBoxes
    return [LiDARBoxes(fx[f"gt_boxes{b}"]) for b in range(bs)], [torch.from_numpy(fx[f"gt_labels{b}"]) for b in range(bs)]


def test_set_prediction_losses_and_decoding_from_the_reference_outputs():
    from distill_bev_amd.center_head import LiDARBoxes
    for name, cfg in (("bevformer_head.npz", "small_bevformer_head_cfg"), ("dgcnn3d_head.npz", "small_dgcnn_head_cfg")):
        head, fx = _head(name, cfg)
        outs = dict(all_cls_scores=torch.from_numpy(fx["all_cls_scores"]), all_bbox_preds=torch.from_numpy(fx["all_bbox_preds"]),
                    enc_cls_scores=None, enc_bbox_preds=None)
        gtb, gtl = _gts(fx, 2)
        losses = head.loss(gtb, gtl, outs)
        assert set(losses) == {"loss_cls", "loss_bbox", "d0.loss_cls", "d0.loss_bbox"}
        for k, v in losses.items():
            ref = float(fx["loss__" + k.replace(".", "_")])
            assert abs(float(v) - ref) <= 1e-5 * abs(ref), (name, k, float(v), ref)
        # `loss` (one read-back for all layers) == the reference's per-layer organisation, value for value
        per_layer = head.loss_reference_order(gtb, gtl, outs)
        assert list(per_layer) == list(losses)
        for k in losses:
            assert abs(float(losses[k]) - float(per_layer[k])) <= 1e-6 * abs(float(per_layer[k])), (name, k)
        dec = head.get_bboxes({k: (v.clone() if torch.is_tensor(v) else v) for k, v in outs.items()},
                              [dict(box_type_3d=lambda t, d=9: LiDARBoxes(t)) for _ in range(2)])
        for b in range(2):
            assert torch.allclose(dec[b][1], torch.from_numpy(fx[f"dec_scores{b}"]), rtol=1e-6, atol=0)
            assert torch.allclose(dec[b][0].tensor, torch.from_numpy(fx[f"dec_boxes{b}"]), rtol=1e-5, atol=1e-6)
            if f"dec_labels{b}" in fx.files:
                assert torch.equal(dec[b][2], torch.from_numpy(fx[f"dec_labels{b}"]))


def test_hungarian_assignment_is_the_minimum_cost_matching():
    """HungarianAssigner3D against brute force over all injections gt -> query on a small case"""
    import itertools
    from distill_bev_amd.detr_head import HungarianAssigner3D, normalize_bbox
    from distill_bev_amd import synthetic as syn
    a = HungarianAssigner3D(cls_cost=dict(type="FocalLossCost", weight=2.0), reg_cost=dict(type="BBox3DL1Cost", weight=0.25),
                            iou_cost=dict(type="IoUCost", weight=0.0), pc_range=[-51.2, -51.2, -5.0, 51.2, 51.2, 3.0])
    g = torch.Generator().manual_seed(2)
    pred, cls = torch.randn((6, 10), generator=g), torch.randn((6, 10), generator=g)
    bx, lb = syn.gt_boxes(3, np.random.default_rng(4))
    from distill_bev_amd.center_head import LiDARBoxes
    gt = torch.cat((LiDARBoxes(bx).gravity_center, torch.from_numpy(bx)[:, 3:]), 1)
    res = a.assign(pred, cls, gt, torch.from_numpy(lb))
    cost = a.cls_cost(cls, torch.from_numpy(lb)) + a.reg_cost(pred[:, :8], normalize_bbox(gt)[:, :8])
    best = min(itertools.permutations(range(6), 3), key=lambda p: sum(float(cost[p[j], j]) for j in range(3)))
    assert [int(q) for q in (res.gt_inds > 0).nonzero().flatten()] == sorted(best)
    for j, q in enumerate(best):
        assert int(res.gt_inds[q]) == j + 1 and int(res.labels[q]) == int(lb[j])
    empty = a.assign(pred, cls, gt[:0], torch.from_numpy(lb)[:0])
    assert bool((empty.gt_inds == 0).all())                       # no ground truth: every query is background


def test_camera_projection_of_the_pillar_points_vs_reference_fixture():
    head, fx = _head()
    enc = head.transformer.encoder
    H, W = [int(v) for v in fx["img_hw"]]
    metas = [dict(lidar2img=list(fx["lidar2img"][b]), img_shape=[(H, W, 3)] * 3) for b in range(2)]
    ref3d = enc.get_reference_points(10, 10, 8.0, 4, dim="3d", bs=2, device="cpu", dtype=torch.float32)
    assert torch.equal(ref3d, torch.from_numpy(fx["ref_3d"]))
    rpc, mask = enc.point_sampling(ref3d, [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], metas)
    m_ref = torch.from_numpy(fx["bev_mask"])
    assert mask.shape == m_ref.shape and int((mask != m_ref).sum()) <= 2
    ok = m_ref & mask
    assert float((rpc - torch.from_numpy(fx["reference_points_cam"])).abs()[ok].max()) <= 1e-5
    ref2d = enc.get_reference_points(10, 10, dim="2d", bs=2, device="cpu", dtype=torch.float32)
    assert ref2d.shape == (2, 100, 1, 2) and abs(float(ref2d[0, 0, 0, 0]) - 0.05) < 1e-7 and abs(float(ref2d[0, 99, 0, 1]) - 0.95) < 1e-7


def test_grid_mask_draws_vs_reference_fixture_on_the_host():
    from distill_bev_amd.bevformer import GridMask
    fx = np.load(os.path.join(GOLD, "grid_mask.npz"))
    gm = GridMask(True, True, rotate=1, offset=False, ratio=0.5, mode=1, prob=0.7).train()
    x = torch.from_numpy(fx["x"])
    for i, seed in enumerate((0, 1, 2, 3, 4, 5)):
        np.random.seed(seed)
        assert torch.equal(gm(x.clone()), torch.from_numpy(fx["y"][i])), seed


def test_previous_bev_rotation_properties():
    """rotate_nearest == torchvision.transforms.functional.rotate (nearest, zero fill) restated: identity at 0 degrees, an
    exact quarter turn about the centre of an odd grid (counter-clockwise for positive angles, as torchvision), zeros
    entering from outside, and composition of +a / -a returning the interior."""
    from distill_bev_amd.transformer import rotate_nearest
    g = torch.Generator().manual_seed(0)
    img = torch.randn((3, 9, 9), generator=g)
    assert torch.equal(rotate_nearest(img, 0.0, center=[4, 4]), img)
    # about the pixel-centre convention of torchvision: centre (4, 4) of a 9 x 9 grid is offset by half a pixel from the grid
    # centre (4.5, 4.5) - 0.5; a quarter turn about the true centre maps the grid onto itself
    q = rotate_nearest(img, 90.0, center=[4.5, 4.5])
    assert torch.equal(q, torch.rot90(img, 1, dims=(1, 2)))
    q = rotate_nearest(img, -90.0, center=[4.5, 4.5])
    assert torch.equal(q, torch.rot90(img, -1, dims=(1, 2)))
    r = rotate_nearest(torch.ones((1, 21, 21)), 45.0, center=[10.5, 10.5])
    assert float(r[0, 0, 0]) == 0.0 and float(r[0, 10, 10]) == 1.0 and 0.5 < float(r.mean()) < 0.9     # corners rotate out
    small = rotate_nearest(rotate_nearest(img, 3.0, center=[4.5, 4.5]), -3.0, center=[4.5, 4.5])
    assert torch.equal(small[:, 2:7, 2:7], img[:, 2:7, 2:7])                                           # sub-pixel turn: nearest keeps cells
