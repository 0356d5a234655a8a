"""graphed.GraphedNoGrad: the gradient-free forward of a conv / norm stack replayed as a hipGraph -- bit-identical to the eager call
step after step while an optimizer moves the weights (packer.WeightPacker re-packs in place), running statistics included; captured
again when module state changes behind it; version counters of the in-place-written statistics move (eval coefficients follow)."""
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


def test_trainer_installs_and_removes_the_adjacent_frame_graph():
    from distill_bev_amd import graphed
    from distill_bev_amd.train_step import Trainer
    src = open(Trainer.__init__.__code__.co_filename).read()
    assert "adjacent_graph" in src and graphed.enabled() in (True, False)
