"""CPU checks of the bench-line bookkeeping added in round 5 (a FLOP-logged kernel was printed as bytes with
frac 17.7): the peak table of the FLOP-logged kernels, the guard that refuses any fraction above 1, the switches of tests/_variants."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_flop_logged_kernels_have_their_own_peak_and_no_fraction_above_one_passes():
    import bench_workloads as W
    assert set(W.FLOP_LOGGED) >= {"wino_fwd", "wino_wgrad", "b6_fwd", "b6_wgrad", "g1_fwd", "g1_wgrad"}
    assert abs(W.FLOP_LOGGED["b6_fwd"][0] - W.MFMA_BF16_PEAK_TF / 6.0) < 1e-9 and W.FLOP_LOGGED["wino_fwd"][0] == W.MFMA_F32_PEAK_TF
    W.assert_fracs({"frac": 0.63, "other": {"b6_fwd": {"frac": 0.35, "frac_of_x": 1.0}, "rows": [1, 2]}})
    with pytest.raises(AssertionError, match="cannot exceed 1"):
        W.assert_fracs({"other_hot_kernels": {"g1_fwd": {"frac": 17.67}}})
    with pytest.raises(AssertionError):
        W.assert_fracs({"frac_of_fp32_mfma_peak": 1.2})
    probe = W._ddp_probe()
    assert probe is None or {"plain_ms", "reducer_overlap_off_ms", "reducer_overlap_on_ms", "source"} <= set(probe)


def test_variant_switches_restore_their_state():
    import _variants as V
    from distill_bev_amd import bn_act, colsum, gemm_bf6, wino
    before = (wino._MIN_WG, gemm_bf6._MIN_ITEMS, gemm_bf6._MIN_WGRAD_ROWS, bn_act._C1["min_rows"], gemm_bf6._ON, colsum._ON,
              bn_act._C1["enabled"], bn_act._state["enabled"], os.environ.get("DBEV_WINO"))
    with V.forced_kernels():
        assert (wino._MIN_WG, gemm_bf6._MIN_ITEMS, gemm_bf6._MIN_WGRAD_ROWS) == (0, 1, 1)
        with V.library_path():
            assert not gemm_bf6._ON and not colsum._ON and not bn_act._state["enabled"] and os.environ["DBEV_WINO"] == "0"
    after = (wino._MIN_WG, gemm_bf6._MIN_ITEMS, gemm_bf6._MIN_WGRAD_ROWS, bn_act._C1["min_rows"], gemm_bf6._ON, colsum._ON,
             bn_act._C1["enabled"], bn_act._state["enabled"], os.environ.get("DBEV_WINO"))
    assert before == after


def test_weight_packer_is_a_no_op_without_device_weights():
    import torch.nn as nn
    from distill_bev_amd.packer import WeightPacker
    p = WeightPacker([nn.Sequential(nn.Conv2d(64, 64, 3, padding=1), nn.Conv2d(64, 64, 1))])     # nothing re-classed, nothing on a GPU
    assert p.wino == [] and p.bf6 == []
    p.repack()
    assert p.launches == 0


def test_shader_clock_sampler_reads_the_starred_level_and_reports_why_when_it_cannot(tmp_path):
    """round 6: bench.py samples the driver's pp_dpm_sclk during the timed region (the current level is the starred line)"""
    import time
    import bench
    f = tmp_path / "pp_dpm_sclk"
    f.write_text("0: 132Mhz\n1: 2174Mhz *\n")
    s = bench._ClockSampler.__new__(bench._ClockSampler)
    import threading
    s.samples, s.source, s.why, s._stop = [], str(f), None, threading.Event()
    s._thread = threading.Thread(target=s._run, daemon=True)
    assert s._read() == 2174.0
    s.start()
    time.sleep(0.12)
    f.write_text("0: 132Mhz *\n1: 2174Mhz\n")
    time.sleep(0.12)
    out = s.stop()
    assert out["samples"] >= 2 and out["max_mhz"] == 2174.0 and out["min_mhz"] == 132.0 and out["nominal_mhz_of_the_peaks"] == 2400.0
    s2 = bench._ClockSampler.__new__(bench._ClockSampler)
    s2.samples, s2.source, s2.why, s2._stop, s2._thread = [], None, "no pp_dpm_sclk file matches the device (0 candidates)", threading.Event(), None
    assert "unavailable" in s2.stop()


def test_packer_reports_what_it_skipped_as_an_empty_list_without_device_weights():
    import torch.nn as nn
    from distill_bev_amd.packer import WeightPacker
    p = WeightPacker([nn.Sequential(nn.Conv2d(64, 64, 1))])
    assert p.repack() == [] and p.skipped == []
