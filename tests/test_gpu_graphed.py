"""graphed.GraphedNoGrad: the gradient-free forward of a conv / norm stack replayed as a hipGraph -- bit-identical to the eager call
step after step while an optimizer moves the weights, running statistics included, whether packer.WeightPacker re-packs in place, skips
a layer, or is absent (the lazy re-pack writes the kept buffers again); after load_state_dict; with an eval-mode norm inside; captured
again when module state changes behind it or a baked-in buffer is no longer the layer's; and the same on the real detector through
Trainer: every replay of the adjacent frame against an eager run of the same modules on the same input and statistics."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _net(dev, seed=3):
    from distill_bev_amd import bn_act, gemm_bf6, wino
    torch.manual_seed(seed)
    net = nn.Sequential(nn.Conv2d(64, 128, 1, bias=False), nn.BatchNorm2d(128), nn.ReLU(),
                        nn.Conv2d(128, 128, 3, padding=1, bias=False), nn.BatchNorm2d(128), nn.ReLU(),
                        nn.Conv2d(128, 64, 1, bias=False), nn.BatchNorm2d(64)).to(dev)
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    bn_act.fuse_bn_relu_modules(net)
    wino.use_wino_convs(net)
    gemm_bf6.use_bf6_convs(net)
    return net.train()


def test_replay_equals_eager_while_the_weights_move(monkeypatch):
    from distill_bev_amd import gemm_bf6, graphed, wino
    from distill_bev_amd.packer import WeightPacker
    monkeypatch.setattr(wino, "_MIN_WG", 0)
    monkeypatch.setattr(gemm_bf6, "_MIN_ITEMS", 1)
    monkeypatch.setattr(gemm_bf6, "_MIN_WGRAD_ROWS", 1)
    monkeypatch.setattr(graphed, "_ON", True)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn((4, 64, 16, 32), generator=g).to(dev).contiguous(memory_format=torch.channels_last) for _ in range(6)]
    results = {}
    for use_graph in (False, True):
        net = _net(dev)
        opt = torch.optim.AdamW(net.parameters(), lr=1e-2, fused=True)
        packer = WeightPacker([net])
        fn = graphed.GraphedNoGrad(lambda t: net(t), token=lambda: graphed.state_token(net),
                                   norms=lambda: [m for m in net.modules() if isinstance(m, nn.BatchNorm2d)]) if use_graph else None
        outs = []
        for x in xs:
            with torch.no_grad():
                y = fn(x) if fn is not None else net(x)
            outs.append(y.clone())
            loss = net(x).square().mean()                      # a training step on the same layers moves the weights
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            packer.repack()
        if fn is not None:
            assert fn.captures == 1 and fn.replays == len(xs) - 2 and fn.eager == 2, (fn.captures, fn.replays, fn.eager)
        results[use_graph] = (outs, [b.clone() for b in net.buffers()])
    for a, b in zip(results[False][0], results[True][0]):
        assert torch.equal(a, b)
    for a, b in zip(results[False][1], results[True][1]):
        assert torch.equal(a, b)                               # running statistics / num_batches_tracked: updated inside the replays


def test_state_change_recaptures_and_eval_coefficients_follow_the_replays(monkeypatch):
    from distill_bev_amd import gemm_bf6, graphed, wino
    monkeypatch.setattr(wino, "_MIN_WG", 0)
    monkeypatch.setattr(gemm_bf6, "_MIN_ITEMS", 1)
    monkeypatch.setattr(graphed, "_ON", True)
    dev = torch.device("cuda:0")
    net = _net(dev, seed=5)
    x = torch.randn((2, 64, 8, 16), device=dev).contiguous(memory_format=torch.channels_last)
    norms = lambda: [m for m in net.modules() if isinstance(m, nn.BatchNorm2d)]
    fn = graphed.GraphedNoGrad(lambda t: net(t), token=lambda: graphed.state_token(net), norms=norms, warmup=1)
    fn(x)
    net.eval()
    with torch.no_grad():
        e0 = net(x).clone()                                    # eval-mode coefficients are now kept on the norm modules
    net.train()
    fn(x)                                                      # captured + replayed: running statistics move inside the graph
    y1 = fn(x).clone()
    assert fn.captures == 1 and fn.replays == 2
    net.eval()                                                 # a training flag the graph baked in: the token changes
    ye = fn(x).clone()
    with torch.no_grad():
        assert torch.equal(ye, net(x))
    ref = _net(dev, seed=5)                                    # the same sequence without the graph: the eval output after the updates
    with torch.no_grad():
        ref.train(); ref(x); ref.eval(); ref(x); ref.train(); ref(x); ref(x); ref.eval()
        assert torch.equal(ye, ref(x)) and not torch.equal(ye, e0)      # not the coefficients kept before the replays
    assert fn.eager >= 2 or fn.captures == 2
    net.train()
    fn(x); fn(x)
    assert fn.captures >= 2
    # while every kernel is being logged the call runs eagerly (bench.py's instrumented steps)
    from distill_bev_amd import _lib as L
    L.kernel_timing(True)
    try:
        r = fn.replays
        fn(x)
        assert fn.replays == r
    finally:
        L.kernel_timing(False)
        L.kernel_timing_read()
    assert y1.shape == ye.shape




def _forced(monkeypatch):
    from distill_bev_amd import gemm_bf6, graphed, wino
    monkeypatch.setattr(wino, "_MIN_WG", 0)
    monkeypatch.setattr(gemm_bf6, "_MIN_ITEMS", 1)
    monkeypatch.setattr(gemm_bf6, "_MIN_WGRAD_ROWS", 1)
    monkeypatch.setattr(graphed, "_ON", True)
    return graphed


def _train_and_compare(monkeypatch, packer_mode, steps=6):
    """the toy stack, `steps` optimizer steps, graph vs no graph; packer_mode: "none" (lazy re-pack only = DBEV_MULTI_PACK=0),
    "skip" (a packer that leaves the first Winograd layer and the first 1x1 layer stale)"""
    graphed = _forced(monkeypatch)
    from distill_bev_amd.packer import WeightPacker
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn((4, 64, 16, 32), generator=g).to(dev).contiguous(memory_format=torch.channels_last) for _ in range(steps)]
    results = {}
    for use_graph in (False, True):
        net = _net(dev)
        opt = torch.optim.AdamW(net.parameters(), lr=1e-2, fused=True)
        packer = WeightPacker([net]) if packer_mode == "skip" else None
        if packer is not None:
            packer.wino, packer.bf6 = packer.wino[1:], packer.bf6[1:]             # these layers are never re-packed by the packer
        fn = graphed.GraphedNoGrad(lambda t: net(t), token=lambda: graphed.state_token(net),
                                   norms=lambda: [m for m in net.modules() if isinstance(m, nn.BatchNorm2d)]) if use_graph else None
        outs, ptrs = [], []
        for x in xs:
            with torch.no_grad():
                y = fn(x) if fn is not None else net(x)
            outs.append(y.clone())
            if fn is not None:
                ptrs.append(sorted(p for _k, _o, ps in fn.baked_in() for p in ps))
            loss = net(x).square().mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            if packer is not None:
                packer.repack()
        if fn is not None:
            assert fn.captures == 1 and fn.dropped == 0 and fn.replays == steps - 2, (fn.captures, fn.dropped, fn.replays)
            assert len(ptrs[-1]) >= 3 and all(p == ptrs[-1] for p in ptrs[2:])    # the baked-in buffers never moved
        results[use_graph] = (outs, [b.clone() for b in net.buffers()])
    for a, b in zip(results[False][0], results[True][0]):
        assert torch.equal(a, b)
    for a, b in zip(results[False][1], results[True][1]):
        assert torch.equal(a, b)


def test_replay_equals_eager_without_a_packer(monkeypatch):
    """DBEV_MULTI_PACK=0 / no WeightPacker at all: before every replay the graph re-validates what it baked in, and the lazy re-pack
    writes the SAME buffers (it used to allocate new ones and the graph kept reading the old)"""
    _train_and_compare(monkeypatch, "none")


def test_replay_equals_eager_when_the_packer_skips_layers(monkeypatch):
    _train_and_compare(monkeypatch, "skip")


def test_multi_pack_switch_off_still_replays_fresh_weights(monkeypatch):
    from distill_bev_amd import packer
    monkeypatch.setattr(packer, "_ON", False)
    graphed = _forced(monkeypatch)
    dev = torch.device("cuda:0")
    net, ref = _net(dev, seed=11), _net(dev, seed=11)
    pk = packer.WeightPacker([net])
    fn = graphed.GraphedNoGrad(lambda t: net(t), token=lambda: graphed.state_token(net), warmup=1)
    x = torch.randn((2, 64, 8, 16), device=dev).contiguous(memory_format=torch.channels_last)
    for i in range(4):
        with torch.no_grad():
            assert torch.equal(fn(x), ref(x))
            for p, q in zip(net.parameters(), ref.parameters()):
                d = torch.randn_like(p) * 0.05
                p.add_(d); q.add_(d)
        assert pk.repack() == []                                   # (switched off: does nothing, reports nothing)
    assert fn.captures == 1 and fn.replays == 3


def test_load_state_dict_after_the_capture(monkeypatch):
    graphed = _forced(monkeypatch)
    dev = torch.device("cuda:0")
    net, other = _net(dev, seed=5), _net(dev, seed=6)
    fn = graphed.GraphedNoGrad(lambda t: net(t), token=lambda: graphed.state_token(net), warmup=1)
    x = torch.randn((2, 64, 8, 16), device=dev).contiguous(memory_format=torch.channels_last)
    fn(x); fn(x); fn(x)
    assert fn.captures == 1 and fn.replays == 2
    net.load_state_dict(other.state_dict())                        # p.copy_: same tensors, versions bumped -- the token does NOT change
    with torch.no_grad():
        y = fn(x).clone()
        assert torch.equal(y, other(x))                            # (both in training mode: batch statistics; `other` has seen nothing else)
    assert fn.captures == 1 and fn.replays == 3 and fn.dropped == 0


def test_a_dropped_cache_drops_the_graph(monkeypatch):
    """a baked-in buffer that is no longer the layer's (here: the caches are deleted, the next use allocates new buffers): the call
    runs eagerly, the graph is captured again after a warm-up -- never a replay against the orphaned buffer"""
    from distill_bev_amd import bn_act
    graphed = _forced(monkeypatch)
    dev = torch.device("cuda:0")
    net, ref = _net(dev, seed=7), _net(dev, seed=7)
    fn = graphed.GraphedNoGrad(lambda t: net(t), token=lambda: graphed.state_token(net), warmup=1)
    x = torch.randn((2, 64, 8, 16), device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(3):
            assert torch.equal(fn(x), ref(x))
        assert fn.captures == 1 and fn.replays == 2
        bn_act.invalidate_eval_coef(net)
        for p, q in zip(net.parameters(), ref.parameters()):
            p.mul_(1.5); q.mul_(1.5)
        for _ in range(3):
            assert torch.equal(fn(x), ref(x))
    assert fn.dropped == 1 and fn.captures == 2 and fn.eager == 2 and fn.replays == 4


def test_eval_mode_norm_inside_the_graph(monkeypatch):
    """a frozen stage (norm in eval mode while the rest trains: mmdet ResNet norm_eval / frozen_stages): its coefficient tensor is
    baked in, kept alive by the graph, not dropped after a replay, and rewritten in place when the norm's
    statistics are written"""
    graphed = _forced(monkeypatch)
    dev = torch.device("cuda:0")
    nets = [_net(dev, seed=9), _net(dev, seed=9)]
    for n in nets:
        n[1].eval()
    net, ref = nets
    norms = lambda: [m for m in net.modules() if isinstance(m, nn.BatchNorm2d)]
    fn = graphed.GraphedNoGrad(lambda t: net(t), token=lambda: graphed.state_token(net), norms=norms, warmup=1)
    x = torch.randn((2, 64, 8, 16), device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for i in range(5):
            junk = torch.empty((1 << 20,), device=dev).normal_()   # allocator churn: a freed coefficient block would be handed out again
            assert torch.equal(fn(x), ref(x)), i
            del junk
            if i == 2:
                for n in nets:
                    n[1].running_mean.add_(0.25); n[1].running_var.mul_(1.5)
    assert "eval_coef" in [k for k, _o, _p in fn.baked_in()]
    assert fn.captures == 1 and fn.dropped == 0 and fn.replays == 4


OPTS = {"model.img_view_transformer.data_config.input_size": (64, 176)}


class _Spy:
    """stands where Trainer put the GraphedNoGrad: every call is answered by the graph AND re-done eagerly on the same input from the
    same running statistics; outputs and statistics afterwards must be bit-equal"""

    def __init__(self, inner, det):
        self.inner, self.det, self.compared = inner, det, 0
        mods = [m for m in (getattr(det, "img_backbone", None), getattr(det, "img_neck", None)) if m is not None]
        self.bufs = [b for m in mods for b in m.buffers()]

    def __call__(self, x):
        before = [b.clone() for b in self.bufs]
        replays = self.inner.replays
        y = self.inner(x)
        if self.inner.replays == replays:
            return y                                               # an eager call (warm-up): nothing to compare
        y = y.clone()
        after = [b.clone() for b in self.bufs]
        for b, v in zip(self.bufs, before):
            b.data.copy_(v)
        with torch.no_grad():
            ye = self.det.image_encoder(x)
        assert torch.equal(ye, y), float((ye - y).abs().max())
        for b, v in zip(self.bufs, after):
            assert torch.equal(b, v)
        self.compared += 1
        return y


@pytest.mark.parametrize("mode", ["packer", "packer_skips_a_stage", "no_packer"])
def test_trainer_replays_equal_eager_on_the_real_detector(monkeypatch, mode):
    import numpy as np
    from _variants import forced_kernels
    from distill_bev_amd import graphed, packer
    from distill_bev_amd.train_step import Trainer, build_model, make_batch
    monkeypatch.setattr(graphed, "_ON", True)
    if mode == "no_packer":
        monkeypatch.setattr(packer, "_ON", False)
    dev = torch.device("cuda:0")
    with forced_kernels():
        model, cfg = build_model(cfg_options=dict(OPTS), seed=3, allow_synthetic_teacher=True)
        tr = Trainer(model, cfg, dev, world_size=1, channels_last=True)
        det = tr.detector
        assert det.adjacent_graph is not None
        if mode == "packer_skips_a_stage":                          # the packer never sees the image backbone's third stage
            gone = {id(p) for p in det.img_backbone.layer3.parameters()}
            tr.packer.wino = [w for w in tr.packer.wino if id(w) not in gone]
            tr.packer.bf6 = [w for w in tr.packer.bf6 if id(w) not in gone]
        spy = _Spy(det.adjacent_graph, det)
        det.adjacent_graph = spy
        rng = np.random.default_rng(5)
        losses = []
        for i in range(5):
            batch = make_batch(1, rng, dev, n_points=8000, input_size=(64, 176))
            loss, _ = tr.step(batch)
            losses.append(float(loss))
            if mode == "packer":
                assert tr.packer.skipped == [], [(tuple(w.shape), why) for w, why in tr.packer.skipped]
        g = spy.inner
        assert g.captures == 1 and g.dropped == 0 and g.replays == 3 and spy.compared == 3, (g.captures, g.dropped, g.replays, spy.compared)
        kinds = {k for k, _o, _p in g.baked_in()}
        assert {"wino_pair", "bf6"} <= kinds, kinds
        assert all(np.isfinite(losses))
        # validation on the same detector: eval mode never takes the graph
        det.eval()
        r = g.replays
        with torch.no_grad():
            det.extract_img_feat(batch["img_inputs"]) if hasattr(det, "extract_img_feat") else None
        assert g.replays == r
        det.train()
        tr.close()
        assert det.adjacent_graph is None
