"""GPU: the fallback ledger (dbev_fallback_note / _count / _reset).  Every host-side mirror of a fused op counts -- and warns
once per reason -- when a device tensor it was wired for takes the stock torch path; eligible calls do not count."""
import warnings

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def test_ineligible_device_calls_are_counted_eligible_ones_are_not():
    from distill_bev_amd import _lib as L
    from distill_bev_amd import bn_act as BA
    from distill_bev_amd.skinny_conv import SkinnyConv2d
    dev = torch.device("cuda:0")
    L.fallback_reset()
    L._warned_fallbacks.clear()            # earlier tests of the same process may have used up the once-per-reason warnings
    bn = nn.BatchNorm2d(64).to(dev).train()
    x = torch.randn(4, 64, 8, 8, device=dev)
    xl = x.contiguous(memory_format=torch.channels_last)
    BA.bn_act(xl, bn, None, True)
    assert L.fallback_counts()["total"] == 0
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        y_ref = BA.bn_act(x, bn, None, True)                 # NCHW: stock torch ops
        BA.bn_act(x, bn, None, True)
        bn24 = nn.BatchNorm2d(24).to(dev).train()            # 6 float4 columns: not a power of two
        BA.bn_act(torch.randn(2, 24, 4, 4, device=dev).contiguous(memory_format=torch.channels_last), bn24, None, False)
    c = L.fallback_counts()
    assert c["bn_act"] == 3 and c["total"] == 3
    msgs = [str(m.message) for m in w if "stock torch path" in str(m.message)]
    assert len(msgs) == 2 and any("not channels-last" in m for m in msgs) and any("24 channels" in m for m in msgs)
    assert torch.allclose(y_ref, BA.bn_act(xl, bn, None, True).contiguous(), atol=1e-5)
    conv = SkinnyConv2d(64, 2, 3, padding=1).to(dev)
    conv(xl)
    assert L.fallback_counts()["skinny_conv"] == 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        conv(x)
    assert L.fallback_counts()["skinny_conv"] == 1 and L.fallback_counts()["total"] == 4
    with BA.disabled():
        BA.bn_act(x, bn, None, True)                         # the parity tests' switch is not a fallback
    assert L.fallback_counts()["total"] == 4
    L.fallback_reset()
    assert L.fallback_counts()["total"] == 0
