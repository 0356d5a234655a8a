"""packer.WeightPacker: every trainable layer's Winograd filters / bf16 planes re-derived in one launch per family after the optimizer
step, into the buffers the layers hold -- bit-equal to what the per-layer (lazy) path packs, and found fresh by the next forward."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _net(dev):
    from distill_bev_amd import gemm_bf6, wino
    torch.manual_seed(4)
    net = nn.Sequential(nn.Conv2d(64, 128, 1, bias=False), nn.ReLU(), nn.Conv2d(128, 128, 3, padding=1, bias=False), nn.ReLU(),
                        nn.Conv2d(128, 64, 1, stride=2, bias=False), nn.ReLU(), nn.Conv2d(64, 64, 3, padding=1, bias=False)).to(dev)
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    assert wino.use_wino_convs(net) == 2 and gemm_bf6.use_bf6_convs(net) == 2
    return net


def test_multi_pack_equals_lazy_packs_and_is_used(monkeypatch):
    from distill_bev_amd import gemm_bf6, wino
    from distill_bev_amd.packer import WeightPacker
    monkeypatch.setattr(wino, "_MIN_WG", 0)
    monkeypatch.setattr(gemm_bf6, "_MIN_ITEMS", 1)
    monkeypatch.setattr(gemm_bf6, "_MIN_WGRAD_ROWS", 1)
    dev = torch.device("cuda:0")
    x = torch.randn((4, 64, 16, 32), device=dev).contiguous(memory_format=torch.channels_last)
    outs = {}
    for use_packer in (False, True):
        net = _net(dev)
        opt = torch.optim.AdamW(net.parameters(), lr=1e-2, fused=True)
        packer = WeightPacker([net]) if use_packer else None
        if packer is not None:
            assert len(packer.wino) == 2 and len(packer.bf6) == 2
        ys = []
        for step in range(3):
            y = net(x)
            opt.zero_grad(set_to_none=True)
            y.square().mean().backward()
            opt.step()
            if packer is not None:
                packer.repack()
                if step == 0:                      # the packs the multi launch wrote == what the single-layer entries write for the new weights
                    for w in packer.wino:      # (a pack buffer holds two format slots, only the one the layer uses is written: compare through the kernel)
                        key, fwd, dgrad, _ = w._dbev_wino_pair
                        assert key[0] == w._version
                        _, _, N, H, W = key
                        Co, C = int(w.shape[0]), int(w.shape[1])
                        xin = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
                        ref = wino.pack_filters(w.detach(), False, (N, C, H, W))
                        assert torch.equal(wino.conv_packed(xin, fwd, Co), wino.conv_packed(xin, ref, Co))
                        if dgrad is not None:
                            gin = torch.randn((N, Co, H, W), device=dev).contiguous(memory_format=torch.channels_last)
                            refd = wino.pack_filters(w.detach(), True, (N, Co, H, W))
                            assert torch.equal(wino.conv_packed(gin, dgrad, C), wino.conv_packed(gin, refd, C))
                    for w in packer.bf6:
                        key, packs, fp = w._dbev_bf6_packs[:3]
                        assert key[0] == w._version and packs
                        for (tr, tn), buf in list(packs.items()):
                            del w._dbev_bf6_packs                     # force a fresh single-layer pack to compare with
                            assert torch.equal(buf, gemm_bf6.packed(w, tr, tn))
                            w._dbev_bf6_packs = (key, packs, fp, {})
            ys.append(y.detach().clone())
        if packer is not None:
            assert packer.launches == 6                               # two families x three steps, nothing left to the lazy path
        outs[use_packer] = ys
    for a, b in zip(outs[False], outs[True]):
        assert torch.equal(a, b)                                      # the training trajectory is bit-identical with and without it
