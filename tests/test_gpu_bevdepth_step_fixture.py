"""The north-star step against the reference's own detector classes.

tests/golden/bevdepth_step.npz (make_golden.py bevdepth_step) holds BEVDepth4DDistill.forward_train of the REFERENCE -- its own
bevdet_distill_more.py / bevdet_distill.py / bevdet.py / centerpoint.py / dynamic_centerpoint.py / mvx_two_stage.py files, built
through their own constructors from the small recipe of tests/golden/standins.py, with the reference's own FPNForBEVDet,
ResNetForBEVDet / BasicBlock (depth net, pre-process net, BEV encoder) and FPN_LSS inside; the one stand-in is the un-vendored
image backbone -- run on CPU: 44 losses, eleven gradients of the total loss along the step and the BEV-encoder weight gradient
split by loss group.  Here the product detector is built from the SAME recipe,
loads the reference's two state dicts strict, and runs the step on the HIP path (lift-splat, DCNv2, dynamic voxelization,
pillar scatter, FG masks, CenterHead targets + loss, FGD terms).

Round 5 adds tests/golden/bevdepth_step_wide.npz (make_golden.py bevdepth_step_wide): the same step of the reference with a BEV encoder
of its own Bottleneck blocks at 64 / 256 channels (standins.distill_cfg(wide=True)) -- channel counts the bf16x6 1x1 GEMMs and the
Winograd 3x3 kernels take -- and two more passes per fixture: `accelerated` = the product as bench.py builds it (channels-last +
accelerate_modules) with the kernels' size thresholds forced to zero; on the wide fixture the accelerated product is also compared
with the plain product directly (losses 2e-5, gradients 1e-4 of their norm; measured 5e-6).

Two passes:
* as is: the 44 losses to 1e-3 (measured 3e-5), the two pooled BEV maps to 1e-3 of their norm (measured 2e-5: the reference
  pools through a cumulative sum over all frustum points, the product sums each cell directly -- DESIGN.md §4).
* aligned: the product's two pooled maps are shifted onto the reference's values (out + (ref - out.detach()): same value, the
  gradient still flows into the product's lift-splat backward), which takes the one operator with a different summation
  order out of the comparison: losses to 1e-4.

Gradients.  The step holds ~1.2 M ReLU gates and L1 signs in the head alone, and (new in round 3) the reference's own training-
mode BatchNorm stacks of depth net / pre-process net / BEV encoder / BEV neck, the deepest over 2 x 4 x 4 = 32 values per channel.
A gate within fp32 rounding of its kink flips between any two implementations (here CPU vs GPU convolutions); under one of the six
heat-map focal terms -- ~100x larger than every other loss at a random initialisation -- one flipped gate moves that term's
gradient by a percent.  So the fixture carries every loss term's own gradient at two small BEV-encoder parameters, and the summed
gradient at eleven parameters along the step both for all 44 losses and for the 38 other than the heat-map terms.
* aligned pass (the comparison proper): >= 34 of the 43 terms agree to 1e-4 of their norm (measured 36; the loose ones: five heat-
  map terms with flipped gates, the worst 1.1e-2), every term to 3e-2; the summed gradient WITHOUT the heat-map terms at image
  backbone / image neck / depth net / DCN / depth head (through the lift-splat backward) / pre-process net / BEV encoder / BEV neck /
  head / adaptation to 5e-4, with them to 2e-2 (measured 5e-3, the flipped gates); FGD groups to 1e-4 / 1e-3; depth-loss path 1e-4.
* as-is pass: the reference's pooled maps carry 2e-5 of cumulative-sum noise, which the 32-value BatchNorms amplify ~100x in the
  gradients -- and differently from run to run (MIOpen's split-K weight-gradient kernels add with float atomics): every term to
  1.5e-1 (measured 4.3e-2 ... 6.3e-2 over several runs), every summed gradient to 1e-1 (measured <= 4.7e-2), the FGD backbone
  group to 1e-3.  The as-is pass says "the same computation"; the aligned pass is the numerical comparison."""
import os
import sys

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
pytestmark = pytest.mark.gpu


def _t(a):
    t = torch.from_numpy(a)
    return t.float() if t.dtype == torch.float16 else t          # (the wide fixture stores fp16-grid weights as float16, exactly)


def _build(fx, dev, wide=False):
    import standins as S
    from distill_bev_amd import detectors  # noqa: F401
    from distill_bev_amd.registry import MODELS, build_detector
    for cls in S.STANDINS.values():
        MODELS.register_module(module=cls, force=True)
    model = build_detector(S.distill_cfg(S.teacher_cfg(), wide=wide))
    sd = {k[7:].replace("__", "."): _t(fx[k]) for k in fx.files if k.startswith("model__")}
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    tsd = {k[9:].replace("__", "."): _t(fx[k]) for k in fx.files if k.startswith("teacher__")}
    model.teacher_model.load_state_dict(tsd, strict=True)
    return model.to(dev).train()


def _groups(losses):
    return {"det": [k for k in losses if k.startswith("task")], "kd_backbone": [k for k in losses if k.endswith("backbone0_backbone2")],
            "kd_head": [k for k in losses if k.endswith("head_head")]}


@pytest.mark.parametrize("mode", ["as_is", "aligned", "accelerated", "wide_aligned", "wide_accelerated"])
def test_bevdepth4d_distill_forward_train_vs_reference_fixture(mode):
    """accelerated (round 5): the aligned pass on the product AS bench.py BUILDS IT -- channels-last weights + `accelerate_modules`
    (train_step.to_channels_last: fused norm + ReLU modules, Winograd / bf16x6 / bias-sum convolution classes, batched head branches,
    folded frozen stacks, HIP upsampling) with the size thresholds of the hand-written convolution kernels forced to zero
    (tests/_variants.forced_kernels), so that every layer of this small recipe the kernels CAN take (the 64-channel 3x3 layers of the
    BEV encoder / BEV neck / teacher SECOND) runs on them, forward and backward; same bounds as the aligned pass, and the kernel event
    log must show the Winograd forward / weight-gradient kernels and the fused norm kernels."""
    import contextlib
    sys.path.insert(0, os.path.dirname(__file__))
    import _variants as V
    wide = mode.startswith("wide_")                 # bevdepth_step_wide.npz: the reference's Bottleneck BEV encoder at 64 / 256 channels,
    mode = mode[5:] if wide else mode               # where the accelerated pass also runs the bf16x6 1x1 GEMMs (forward, both gradients,
    aligned, accelerated = mode != "as_is", mode == "accelerated"         # statistics epilogues) -- standins.distill_cfg(wide=True)
    with contextlib.ExitStack() as stack:
        if accelerated:
            stack.enter_context(V.forced_kernels())
            from distill_bev_amd import _lib as L
            L.kernel_timing_read()
            L.kernel_timing(True)
            stack.callback(L.kernel_timing, False)
        res = _fixture_pass(aligned, accelerated, wide)
        if accelerated:
            ran = {k: len(v) for k, v in L.kernel_timing_read().items()}
            print("kernels", ran)
            assert ran.get("wino_fwd", 0) >= 4 and ran.get("wino_wgrad", 0) >= 1, ran
            assert any(k.startswith("bn_apply") for k in ran) and any(k.startswith("bn_bwd_dx") for k in ran), ran
            if wide:
                assert ran.get("b6_fwd", 0) >= 8 and ran.get("b6_wgrad", 0) >= 4, ran
    if accelerated and wide:
        # ... and the accelerated product against the plain product (library convolutions, torch norms) on the same fixture: the hand-written
        # kernels move NOTHING beyond fp32 round-off -- losses to 2e-5, every gradient along the step to 1e-4 of its norm
        ref = _fixture_pass(True, False, True)
        for k in ref["losses"]:
            assert abs(res["losses"][k] - ref["losses"][k]) <= 2e-5 * abs(ref["losses"][k]) + 1e-7, (k, res["losses"][k], ref["losses"][k])
        worst = 0.0
        for n, g in ref["grad"].items():
            e = float((res["grad"][n] - g).norm() / g.norm().clamp_min(1e-20))
            worst = max(worst, e)
            assert e <= 1e-4, (n, e)
        print("accelerated vs plain product: worst gradient rel-L2", worst)


def _fixture_pass(aligned, accelerated, wide=False):
    from distill_bev_amd.center_head import LiDARBoxes
    fx = np.load(os.path.join(GOLD, "bevdepth_step_wide.npz" if wide else "bevdepth_step.npz"))
    dev = torch.device("cuda:0")
    model = _build(fx, dev, wide)
    assert model.training and not model.teacher_model.training
    if accelerated:
        from distill_bev_amd.train_step import to_channels_last
        n_bn, n_up = to_channels_last(model)
        print("rewired: fused norm pairs", n_bn, "upsample", n_up, "wino", model.wino_convs, "folded", model.folded_norm_pairs,
              "bf6", model.bf6_convs, "bias-sum", model.bias_sum_convs, "batched branches", model.batched_branches,
              "skinny", model.skinny_convs)
        assert n_bn >= 10 and model.wino_convs >= 4 and model.bias_sum_convs >= 4
    vt = model.img_view_transformer
    splat, calls = vt.lift_splat_cameras, []

    def lift_splat_cameras(*a, **k):
        out = splat(*a, **k)
        ref = torch.from_numpy(fx[f"pooled{len(calls)}"]).to(dev)
        calls.append(float((out.detach() - ref).norm() / ref.norm()))
        return out + (ref - out.detach()) if aligned else out
    vt.lift_splat_cameras = lift_splat_cameras
    B = 2
    img_inputs = tuple(torch.from_numpy(fx[k].astype(np.float32)).to(dev)
                       for k in ("imgs", "rots", "trans", "intrins", "post_rots", "post_trans", "depth_gt"))
    points = [torch.from_numpy(fx[f"points{b}"]).to(dev) for b in range(B)]
    gtb = [LiDARBoxes(fx[f"gt_boxes{b}"]) for b in range(B)]
    gtl = [torch.from_numpy(fx[f"gt_labels{b}"]).to(dev) for b in range(B)]
    losses = model.forward_train(points=points, img_metas=None, gt_bboxes_3d=gtb, gt_labels_3d=gtl, img_inputs=img_inputs)
    # the pooled maps themselves: the product against the reference's cumulative-sum pooling
    assert len(calls) == 2 and max(calls) <= 1e-3, calls
    want = {k[6:]: float(fx[k]) for k in fx.files if k.startswith("loss__")}
    got = {k.replace(".", "_"): float(v.detach()) for k, v in losses.items()}
    assert set(got) == set(want), sorted(set(got) ^ set(want))
    assert len(got) == 44
    ltol = 1e-4 if aligned else 1e-3
    worst = max(got, key=lambda k: abs(got[k] - want[k]) / max(abs(want[k]), 1e-6))
    print("pooled rel2", calls, "worst loss:", worst, got[worst], want[worst])
    for k in want:
        assert abs(got[k] - want[k]) <= ltol * abs(want[k]) + 1e-6, (k, got[k], want[k])
    params = dict(model.named_parameters())
    rel = lambda g, ref: float((g.cpu() - ref).norm() / ref.norm().clamp_min(1e-20))

    # every term's own gradient at two small BEV-encoder parameters
    tb = [params["img_bev_encoder_backbone.layers.0.0.bn1.bias"], params["img_bev_encoder_neck.conv.1.bias"]]
    errs = {}
    for k, v in losses.items():
        key = "term__" + k.replace(".", "_")
        if key not in fx.files:
            assert k == "loss_depth"
            continue
        gr = torch.autograd.grad(v, tb, retain_graph=True, allow_unused=True)
        g = torch.cat([(t if t is not None else torch.zeros_like(p)).reshape(-1) for t, p in zip(gr, tb)])
        errs[k] = rel(g, torch.from_numpy(fx[key]))
    out = {"grad": {}, "gradnh": {}, "terms": {}}
    tight = [k for k, e in errs.items() if e <= 1e-4]
    print("terms", len(errs), "tight", len(tight), "loose", {k: e for k, e in errs.items() if e > 1e-4})
    term_report = (len(errs), len(tight), max(errs.values()))

    # summed gradients at eleven parameters along the step (all losses / all but the heat-map terms), and the BEV-encoder weight
    # gradient by loss group
    bad = []
    names = [k[6:].replace("__", ".") for k in fx.files if k.startswith("grad__")]
    assert len(names) == (15 if wide else 11)
    grads = torch.autograd.grad(sum(losses.values()), [params[n] for n in names], retain_graph=True)
    smooth = sum(v for k, v in losses.items() if not k.endswith("loss_heatmap"))
    grads_nh = torch.autograd.grad(smooth, [params[n] for n in names], retain_graph=True, allow_unused=True)
    for n, g, gn in zip(names, grads, grads_nh):
        e = rel(g, torch.from_numpy(fx["grad__" + n.replace(".", "__")]))
        key = "gradnh__" + n.replace(".", "__")
        en = rel(gn, torch.from_numpy(fx[key])) if key in fx.files else 0.0
        print(n, e, en)
        out["grad"][n], out["gradnh"][n] = g.detach().cpu(), (None if gn is None else gn.detach().cpu())
        if e > (2e-2 if aligned else 1e-1) or en > ((2e-3 if wide else 5e-4) if aligned else 1e-1):
            bad.append((n, e, en))
    w0 = params["img_bev_encoder_backbone.layers.0.0.conv1.weight"]
    gbar = dict(det=2e-2, kd_backbone=1e-4, kd_head=1e-3) if aligned else dict(det=1e-1, kd_backbone=1e-3, kd_head=1e-1)
    if wide:
        gbar = dict(det=2e-2, kd_backbone=1e-3, kd_head=1e-2)
    for gname, keys in _groups(losses).items():
        g = torch.autograd.grad(sum(losses[k] for k in keys), w0, retain_graph=True)[0]
        e = rel(g, torch.from_numpy(fx["gradgroup__" + gname]))
        print("group", gname, len(keys), e)
        if e > gbar[gname]:
            bad.append((gname, e))
    # the depth loss alone (sigmoid + BCE on the depth logits): its whole backward, image backbone included
    n = "img_view_transformer.depthnet.bias"
    g = torch.autograd.grad(losses["loss_depth"], params[n], retain_graph=True)[0]
    e = rel(g, torch.from_numpy(fx["grad_depth__depthnet_bias"]))
    print("depth", e)
    if e > 1e-4:
        bad.append(("depth", e))
    print("REPORT", aligned, term_report, bad)
    if aligned and wide:
        # the wide recipe's Bottleneck stack (norms over 128-512 values behind 64-channel 1x1 layers) amplifies the CPU-vs-GPU rounding
        # of the convolutions in front of it ~10x more than the narrow one: every term to 3e-2 (measured 1.8e-4 ... 1.6e-2, the SAME
        # figures to four digits for the plain and the accelerated product -- which are compared with each other directly below)
        assert term_report[0] == 43 and term_report[2] <= 3e-2, (term_report, errs)
    elif aligned:
        assert term_report[0] == 43 and term_report[1] >= 34 and term_report[2] <= 3e-2, (term_report, errs)
    else:
        assert term_report[0] == 43 and term_report[2] <= 1.5e-1, (term_report, errs)
    assert not bad, bad
    out["losses"] = got
    return out
