"""The other detector names of the reference's config surface (BEVDepth4D, BEVDepthDistill, BEVDetDistill,
BEVDet4DDistill) built through the registry from variations of the recipe and run for one training forward/backward
at reduced image size."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SIZE = (64, 176)


def _cfg():
    from distill_bev_amd.config import Config
    from distill_bev_amd.train_step import DEFAULT_CONFIG
    m = copy.deepcopy(Config.fromfile(DEFAULT_CONFIG).model)
    m = m.to_dict() if hasattr(m, "to_dict") else dict(m)
    m["img_view_transformer"]["data_config"] = dict(m["img_view_transformer"]["data_config"]); m["img_view_transformer"]["data_config"]["input_size"] = SIZE
    return m


def _build(m):
    from distill_bev_amd.registry import build_detector
    import distill_bev_amd.detectors  # noqa: F401
    if m.get("teacher_config") is not None and m.get("inherit_head") and not m.get("teacher_ckpt"):
        from distill_bev_amd.train_step import synthetic_teacher_checkpoint      # the recipe asserts a teacher checkpoint
        m["teacher_ckpt"] = synthetic_teacher_checkpoint(m, 0)
    torch.manual_seed(0)
    det = build_detector(m)
    det.init_weights()
    return det.to("cuda:0").train()


def _batch(single_frame):
    from distill_bev_amd.train_step import make_batch
    b = make_batch(2, np.random.default_rng(3), torch.device("cuda:0"), n_points=20000, input_size=SIZE)
    if single_frame:
        imgs, rots, trans, intr, prots, ptrans, depth = b["img_inputs"]
        B = imgs.shape[0]
        h, w = imgs.shape[-2:]
        imgs = imgs.view(B, 6, 2, 3, h, w)[:, :, 0].contiguous()
        first = lambda t: t.view(B, 2, 6, *t.shape[2:])[:, 0].contiguous()
        b["img_inputs"] = (imgs, first(rots), first(trans), first(intr), first(prots), first(ptrans), first(depth))
    return b


def _finite_backward(det, losses):
    from distill_bev_amd.train_step import parse_losses
    assert all(torch.isfinite(v) for v in losses.values())
    parse_losses(losses).backward()
    g = [p.grad for p in det.parameters() if p.requires_grad]
    assert all(x is not None and torch.isfinite(x).all() for x in g), "every student parameter takes part"


def test_bevdepth4d_is_the_student_without_the_distillation_terms():
    m = _cfg()
    full = _build(copy.deepcopy(m))
    for k in ("teacher_config", "teacher_ckpt", "distill_type", "distill_params", "eval_teacher", "inherit_head"):
        m.pop(k, None)
    m["type"] = "BEVDepth4D"
    plain = _build(m)
    assert plain.teacher_model is None
    missing, unexpected = plain.load_state_dict(full.state_dict(), strict=False)
    assert not missing                                            # the student's keys are a subset of the distill model's
    assert all(k.split(".")[0] in ("channel_wise_adaptations", "spatial_wise_adaptations", "teacher_adaptations")
               for k in unexpected)
    b = _batch(False)
    lp = plain.forward_train(**b)
    lf = full.forward_train(**b)
    assert set(lp) < set(lf) and len(lp) == 37                    # depth + 6 tasks x 6 terms
    for k in lp:
        # two model instances in one process: MIOpen may serve the second one with a different (by then tuned) solver
        # for the same convolution; the random-init network amplifies that round-off (same bound as the step parity test)
        assert abs(float(lp[k].detach()) - float(lf[k].detach())) <= 3e-3 * max(abs(float(lf[k].detach())), 1e-3), k
    _finite_backward(plain, lp)


@pytest.mark.parametrize("name", ["BEVDepthDistill", "BEVDetDistill", "BEVDet4DDistill"])
def test_distillation_variants_train_one_step(name):
    m = _cfg()
    m["type"] = name
    two_frames = name == "BEVDet4DDistill"
    if not two_frames:                                            # one frame: no adjacent BEV feature is concatenated
        for k in ("aligned", "detach", "before", "pre_process"):
            m.pop(k, None)
        m["img_bev_encoder_backbone"]["numC_input"] = 64
    if name != "BEVDepthDistill":                                 # plain LSS view transformer: no depth head
        vt = m["img_view_transformer"]
        m["img_view_transformer"] = dict(type="ViewTransformerLiftSplatShoot", grid_config=vt["grid_config"],
                                         data_config=vt["data_config"], numC_input=512, numC_Trans=vt["numC_Trans"])
    det = _build(m)
    losses = det.forward_train(**_batch(not two_frames))
    has_depth = name == "BEVDepthDistill"
    assert ("loss_depth" in losses) == has_depth
    assert sum(k.startswith("kd_") for k in losses) == 10         # 3 positions x (fg, bg, spatial) + fp term at the head
    assert len(losses) == 36 + 10 + int(has_depth)
    _finite_backward(det, losses)


def test_skipping_the_unread_teacher_branches_changes_no_loss(monkeypatch):
    """forward_distill evaluates only the teacher's heat-map branches (the one thing add_fp_as_fg reads); with
    DBEV_TEACHER_FULL_HEAD=1 all 36 branch stacks run as in the reference -- the same losses."""
    model = _build(_cfg()).eval()                 # eval: BatchNorm on running statistics -> two forwards see the same state
    batch = _batch(False)
    with torch.no_grad():
        a = model.forward_train(**batch)
        monkeypatch.setenv("DBEV_TEACHER_FULL_HEAD", "1")
        b = model.forward_train(**batch)
    assert set(a) == set(b) and any("kd_fp_bg_feat_loss" in k for k in a)
    for k in a:        # (two forwards of the image branch are not bit-identical by themselves: MIOpen's split-K convolutions)
        assert abs(float(a[k]) - float(b[k])) <= 1e-5 * max(abs(float(b[k])), 1e-6), k
