"""CPU: the oracle's restatements against fixtures computed by the reference's OWN files imported by path
(tests/golden/make_golden.py sections fgd / shift_depth / centerloss / pfn / second; stubs in _ref_import.py):

  fgd_losses.npz    BEVDetDistill.fgd_distill_loss / foreground_scale_mask / add_fp_as_fg   (bevdet_distill.py)
  shift_depth.npz   BEVDet4DDistill.shift_feature, BEVDepthDistill.get_depth_loss           (bevdet_distill_more.py)
  center_loss.npz   CenterHead.get_targets + CenterHead.loss                                (centerpoint_head.py)
  pfn_scatter.npz   _dynamic_scatter fwd/bwd, DynamicPillarFeatureNet.forward               (scatter_points.py, pillar_encoder.py)
  second_fpn.npz    SECOND + SECONDFPN forward                                              (second.py, second_fpn.py)

This is what pins oracle/distill.py (loss half), oracle/step_ops.py, oracle/center_targets.py and the dynamic scatter of
oracle/voxel.c; the -m gpu tests then hold the HIP path to the same fixtures (tests/test_gpu_reference_fixtures.py).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import center_targets as OCT
from oracle import distill as OD
from oracle import step_ops as OS
from oracle import voxel as OV


def _rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-12)


# ---- FGD ------------------------------------------------------------------------------------------
def _fgd_inputs(g):
    boxes = [g["boxes0"], g["boxes1"]]
    fg, fs, bs = OD.foreground_scale_mask(32, 32, boxes, tuple(g["grid_size"]), tuple(g["pc_range"]), tuple(g["voxel_size"]))
    return boxes, fg, fs, bs


def test_foreground_masks_of_the_fgd_fixture():
    g = load_golden("fgd_losses.npz")
    _, fg, fs, bs = _fgd_inputs(g)
    assert np.array_equal(fg, g["head_fg"]) and np.array_equal(bs, g["head_bg_scale"])
    assert np.allclose(fs, g["head_fg_scale"], rtol=2.5e-7, atol=0)       # torch.sqrt on the CPU: 1 ulp
    assert fg.sum() > 20


def test_fp_mask_restatements_match_reference_add_fp_as_fg():
    g = load_golden("fgd_losses.npz")
    _, fg, _, _ = _fgd_inputs(g)
    edges = np.cumsum([0] + list(g["ncls"]))
    split = lambda a: [torch.from_numpy(a[:, edges[i]:edges[i + 1]].copy()) for i in range(len(edges) - 1)]
    fp, sc, n = OS.add_fp_as_fg("teacher", torch.from_numpy(fg), split(g["gt_hm"]), split(g["t_logit"]), split(g["s_sig"]), 0.1)
    assert np.array_equal(fp.numpy(), g["head_fp"]) and np.array_equal(n.numpy(), g["head_n_fp"])
    assert np.array_equal(sc.numpy(), g["head_fp_scale"])
    assert n.min() > 0
    # numpy twin used by the fp64 loss yardstick
    t_max = 1 / (1 + np.exp(-g["t_logit"].astype(np.float64)))
    t_max = np.clip(t_max.astype(np.float32), 1e-4, 1 - 1e-4).max(1, keepdims=True)
    fp2, sc2, n2 = OD.fp_masks(fg, g["gt_hm"].max(1, keepdims=True), t_max, 0.1)
    assert np.array_equal(fp2, g["head_fp"]) and np.array_equal(n2, g["head_n_fp"])


def test_fgd_head_position_losses_match_reference():
    """1x1-conv adaptation -> attention masks -> combine_gt scaling -> fg / bg / fp sums + spatial term."""
    g = load_golden("fgd_losses.npz")
    _, fg, fs, bs = _fgd_inputs(g)
    S = OD.conv1x1(g["head_student_in"], g["head_adapt__weight"], g["head_adapt__bias"])
    T = g["head_teacher"]
    out, aux = OD.fgd_feature_losses(S, T, fg, fs, bs, fp=g["head_fp"].astype(np.float64),
                                     fp_scale=g["head_fp_scale"].astype(np.float64), n_fp=g["head_n_fp"].astype(np.float64))
    out["kd_spatial_loss"] = OD.spatial_loss(T, S, g["head_spat__weight"], g["head_spat__bias"], 2.5e-3)
    for k in ("kd_fg_feat_loss", "kd_bg_feat_loss", "kd_fp_bg_feat_loss", "kd_spatial_loss"):
        assert _rel(out[k], g[f"head_loss__{k}"]) < 2e-6, (k, out[k], g[f"head_loss__{k}"])


def test_fgd_backbone_position_losses_match_reference():
    """Upsample x4 (bilinear, align_corners) + ThreeLayer with training BatchNorm, no fp term."""
    g = load_golden("fgd_losses.npz")
    _, fg, fs, bs = _fgd_inputs(g)
    sd = {k[len("backbone_adapt__1__"):]: v for k, v in g.items() if k.startswith("backbone_adapt__1__")}
    S = OD.three_layer(OD.upsample_bilinear_ac(g["backbone_student_in"], 4), sd)
    T = g["backbone_teacher"]
    out, _ = OD.fgd_feature_losses(S, T, fg, fs, bs)
    out["kd_spatial_loss"] = OD.spatial_loss(T, S, g["backbone_spat__weight"], g["backbone_spat__bias"], 2.5e-3)
    for k in ("kd_fg_feat_loss", "kd_bg_feat_loss", "kd_spatial_loss"):
        assert _rel(out[k], g[f"backbone_loss__{k}"]) < 5e-6, (k, out[k], g[f"backbone_loss__{k}"])
    assert "backbone_loss__kd_fp_bg_feat_loss" not in g


# ---- shift_feature / depth loss -----------------------------------------------------------------------
def test_shift_feature_matches_reference():
    g = load_golden("shift_depth.npz")
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    trans = [torch.from_numpy(g["trans0"]), torch.from_numpy(g["trans1"])]
    rots = [torch.from_numpy(g["rots0"]), torch.from_numpy(g["rots1"])]
    y = OS.shift_feature(x, trans, rots, torch.from_numpy(g["dx"]), torch.from_numpy(g["bx"]))
    assert np.abs(y.detach().numpy() - g["shift_bilinear"]).max() < 1e-6
    (gx,) = torch.autograd.grad(y, x, torch.from_numpy(g["shift_grad_out"]))
    assert np.abs(gx.numpy() - g["shift_grad_in"]).max() < 1e-5
    yn = OS.shift_feature(x.detach(), trans, rots, torch.from_numpy(g["dx"]), torch.from_numpy(g["bx"]), "nearest")
    assert (yn.numpy() != g["shift_nearest"]).mean() < 1e-3
    assert np.abs(g["shift_bilinear"] - g["x"]).max() > 0.1          # the warp is not the identity


def test_depth_loss_matches_reference():
    g = load_golden("shift_depth.npz")
    logits = torch.from_numpy(g["depth_logits"]).requires_grad_(True)
    loss = OS.get_depth_loss(torch.from_numpy(g["depth_gt"]), logits, [1.0, 8.0, 1.0], 7, 100.0)
    assert _rel(loss, g["loss_depth"]) < 1e-6
    (gl,) = torch.autograd.grad(loss, logits)
    assert np.abs(gl.numpy() - g["grad_logits"]).max() < 1e-6


# ---- CenterHead -----------------------------------------------------------------------------------------
class _Head:
    """what oracle/center_targets.py reads from a CenterHead"""
    def __init__(self, tasks, cfg, n):
        self.class_names, self.train_cfg, self.task_heads, self.norm_bbox = tasks, cfg, [None] * n, True


CENTER_TASKS = [["car"], ["truck", "construction_vehicle"], ["bus", "trailer"], ["barrier"], ["motorcycle", "bicycle"],
                ["pedestrian", "traffic_cone"]]
CENTER_CFG = dict(grid_size=[256, 256, 40], point_cloud_range=[-12.8, -12.8, -5.0, 12.8, 12.8, 3.0],
                  voxel_size=[0.1, 0.1, 0.2], out_size_factor=8, dense_reg=1, gaussian_overlap=0.1, max_objs=12,
                  min_radius=2, code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2])


def center_fixture_inputs(g, device="cpu"):
    from distill_bev_amd.center_head import LiDARBoxes
    boxes = [LiDARBoxes(g["boxes0"]), LiDARBoxes(g["boxes1"])]
    labels = [torch.from_numpy(g["labels0"]), torch.from_numpy(g["labels1"])]
    preds, leaves = [], []
    for t in range(6):
        d = {}
        for k in ("reg", "height", "dim", "rot", "vel", "heatmap"):
            leaf = torch.from_numpy(g[f"pred{t}_{k}"]).to(device).requires_grad_(True)
            leaves.append((f"pred{t}_{k}", leaf))
            d[k] = leaf * 1.0
        preds.append([d])
    return boxes, labels, preds, leaves


def test_center_targets_and_loss_match_reference_centerhead():
    g = load_golden("center_loss.npz")
    boxes, labels, preds, leaves = center_fixture_inputs(g)
    head = _Head(CENTER_TASKS, CENTER_CFG, 6)
    hms, annos, inds, masks = OCT.get_targets(head, boxes, labels, torch.device("cpu"))
    assert np.array_equal(torch.stack(masks).numpy(), g["mask"]) and np.array_equal(torch.stack(inds).numpy(), g["ind"])
    assert np.array_equal(torch.cat(hms, 1).numpy(), g["heatmap"])
    a, r = torch.stack(annos).numpy(), g["anno_box"]
    assert np.abs(a - r).max() < 1e-6
    assert g["mask"][0, 0].sum() == 11        # 12 slots, one of them a skipped out-of-range box
    losses = OS.centerhead_loss(preds, hms, annos, inds, masks, CENTER_CFG["code_weights"])
    keys = [k for k in g if k.startswith("loss__")]
    assert len(keys) == 36
    for k in keys:
        name = k[len("loss__"):].replace("__", ".")
        assert _rel(losses[name], g[k]) < 2e-6, (name, float(losses[name]), float(g[k]))
    grads = torch.autograd.grad(sum(losses.values()), [l for _, l in leaves])
    for (n, _), gr in zip(leaves, grads):
        ref = g["grad_" + n]
        assert np.abs(gr.numpy() - ref).max() <= 2e-6 * max(np.abs(ref).max(), 1.0), n


# ---- dynamic scatter / PFN --------------------------------------------------------------------------------
@pytest.mark.parametrize("red", ["max", "mean", "sum"])
def test_dynamic_scatter_c_oracle_matches_reference_python_op(red):
    g = load_golden("pfn_scatter.npz")
    vf, vc, cmap, cnt = OV.dynamic_scatter_forward(g["ds_feats"], g["ds_coors"], red)
    assert np.array_equal(vc, g[f"ds_{red}_coors"])
    if red == "max":
        assert np.array_equal(vf, g["ds_max_feats"])
    else:
        assert np.abs(vf - g[f"ds_{red}_feats"]).max() < 2e-6
    gin = OV.dynamic_scatter_backward(g[f"ds_{red}_gout"], g["ds_feats"], vf, cmap, cnt, red)
    if red == "mean":
        assert np.abs(gin - g["ds_mean_gin"]).max() < 1e-6
    else:
        assert np.array_equal(gin, g[f"ds_{red}_gin"])           # max: arg-max ties -> lowest point index
    assert (g["ds_coors"] < 0).any(1).sum() > 30


@pytest.mark.parametrize("training", [True, False])
def test_pillar_feature_net_oracle_matches_reference_module(training):
    g = load_golden("pfn_scatter.npz")
    pre = "pfn_sd__pfn_layers__0__" if training else "pfn_eval_sd__pfn_layers__0__"
    t = lambda k: torch.from_numpy(g[pre + k])
    vf, vc = OS.pillar_feature_net(torch.from_numpy(g["pfn_points"]), torch.from_numpy(g["pfn_coors"]), t("0__weight"),
                                   t("1__weight"), t("1__bias"), t("1__running_mean"), t("1__running_var"), 1e-3,
                                   g["voxel_size"], g["pc_range"], training)
    assert np.array_equal(vc, g["pfn_voxel_coors"])
    ref = g["pfn_voxel_feats"] if training else g["pfn_eval_voxel_feats"]
    assert np.abs(vf - ref).max() < 2e-5
    if not training:
        canvas = OV.pillars_scatter(vf, vc, 2, 16, 16)
        assert np.abs(canvas - g["pfn_eval_canvas"]).max() < 2e-5


# ---- SECOND / SECONDFPN (product torch modules, same weights) -----------------------------------------
def second_fixture_modules(g):
    from distill_bev_amd import nets  # noqa: F401  (registers SECOND / SECONDFPN)
    from distill_bev_amd.registry import build_backbone, build_neck
    bb = build_backbone(dict(type="SECOND", in_channels=8, out_channels=[8, 16, 32], layer_nums=[1, 2, 2],
                             layer_strides=[2, 2, 2], norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01),
                             conv_cfg=dict(type="Conv2d", bias=False)))
    nk = build_neck(dict(type="SECONDFPN", in_channels=[8, 16, 32], out_channels=[8, 8, 8], upsample_strides=[0.5, 1, 2],
                         norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), upsample_cfg=dict(type="deconv", bias=False),
                         use_conv_for_no_stride=True))
    for mod, pre in ((bb, "bb__"), (nk, "nk__")):
        sd = {k[len(pre):].replace("__", "."): torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith(pre)}
        missing, unexpected = mod.load_state_dict(sd, strict=True)
        assert not missing and not unexpected
    return bb.eval(), nk.eval()


def test_second_and_fpn_same_weights_outputs_match_reference_modules():
    """state-dict keys of the product's SECOND / SECONDFPN equal the reference's (strict load) and so do the outputs"""
    g = load_golden("second_fpn.npz")
    bb, nk = second_fixture_modules(g)
    with torch.no_grad():
        feats = bb(torch.from_numpy(g["x"]))
        y = nk(feats)
    for i, f in enumerate(feats):
        assert np.abs(f.numpy() - g[f"f{i}"]).max() < 2e-5
    assert np.abs(y[0].numpy() - g["y"]).max() < 2e-5


# ---- the whole-step oracle's own FGD sequence (oracle/cpu_step.py) --------------------------------------
@pytest.mark.parametrize("tag", ["head", "backbone"])
def test_cpu_step_fgd_sequence_matches_reference(tag):
    """CpuBEVDepth4DDistill.fgd_distill_loss (the torch op sequence bench.py times as cpu_baseline and the step-parity
    test compares the HIP path with) on a bare instance vs the imported reference's losses and gradients."""
    import torch.nn as nn
    from distill_bev_amd import detectors as D
    from distill_bev_amd.center_head import L1Loss, LiDARBoxes, MSELoss
    from oracle.cpu_step import CpuBEVDepth4DDistill
    from types import SimpleNamespace
    g = load_golden("fgd_losses.npz")
    det = CpuBEVDepth4DDistill.__new__(CpuBEVDepth4DDistill)
    nn.Module.__init__(det)
    object.__setattr__(det, "teacher_model", None)
    fp = "teacher" if tag == "head" else "none"
    det.distill_params = dict(
        spatial_t=0.5, spatial_student_ratio=1.0, channel_t=0.5, fg_feat_loss_weights=[6e-3], bg_feat_loss_weights=[4e-2],
        spatial_loss_weights=[2.5e-3], spatial_attentions=["teacher_student"], spatial_mask=True, fp_as_foreground=[fp],
        output_threshold=0.1, groundtruth_threshold=None, fp_weight=6e-2, fp_epoch=0)
    det._epoch = 1
    if tag == "head":
        adapt = nn.Conv2d(12, 16, kernel_size=1)
        adapt.load_state_dict({"weight": torch.from_numpy(g["head_adapt__weight"]), "bias": torch.from_numpy(g["head_adapt__bias"])})
    else:
        adapt = nn.Sequential(nn.Upsample(scale_factor=4, mode="bilinear", align_corners=True),
                              D.ThreeLayer(in_features=6, out_features=8, kernel_size=1, stride=1))
        pre = "backbone_adapt__"
        adapt.load_state_dict({k[len(pre):].replace("__", "."): torch.from_numpy(np.asarray(v)) for k, v in g.items()
                               if k.startswith(pre)}, strict=True)
    spat = nn.Conv2d(1, 1, kernel_size=3, padding=1)
    spat.load_state_dict({"weight": torch.from_numpy(g[f"{tag}_spat__weight"]), "bias": torch.from_numpy(g[f"{tag}_spat__bias"])})
    det.teacher_adaptations = nn.ModuleList([nn.Identity()])
    det.channel_wise_adaptations = nn.ModuleList([adapt])
    det.spatial_wise_adaptations = nn.ModuleList([spat])
    det.feat_criterion, det.spatial_criterion = MSELoss(reduction="none"), L1Loss(reduction="none")
    object.__setattr__(det, "pts_bbox_head", SimpleNamespace(train_cfg=dict(
        grid_size=[int(v) for v in g["grid_size"]], point_cloud_range=[float(v) for v in g["pc_range"]],
        voxel_size=[float(v) for v in g["voxel_size"]])))
    det.train()
    edges = np.cumsum([0] + list(g["ncls"]))
    split = lambda a: [torch.from_numpy(a[:, edges[i]:edges[i + 1]].copy()) for i in range(len(edges) - 1)]
    s_in = torch.from_numpy(g[f"{tag}_student_in"]).requires_grad_(True)
    losses = det.fgd_distill_loss(torch.from_numpy(g[f"{tag}_teacher"]), s_in, [LiDARBoxes(g["boxes0"]), LiDARBoxes(g["boxes1"])],
                                  None, None, split(g["gt_hm"]), [[dict(heatmap=t)] for t in split(g["t_logit"])],
                                  [[dict(heatmap=s)] for s in split(g["s_sig"])], 0)
    keys = [k[len(tag) + 7:] for k in g if k.startswith(f"{tag}_loss__")]
    assert set(keys) == set(losses)
    for k in keys:
        assert _rel(losses[k], g[f"{tag}_loss__{k}"]) < 2e-5, (k, float(losses[k]), float(g[f"{tag}_loss__{k}"]))
    (gs,) = torch.autograd.grad(sum(losses.values()), s_in)
    ref = g[f"{tag}_grad_student_in"]
    assert np.abs(gs.numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
