"""Switches shared by the detector-level GPU tests: the SAME built detector run (a) with every hand-written dense kernel forced on
whatever the layer's size (the thresholds that keep small layers on the library are about speed, not correctness) and (b) on the
library's kernels with the unfused norm / activation op sequence -- the reference's own module sequence
(mmdet3d/models/detectors/bevdet_distill_more.py:457-522 calls plain nn.Conv2d / nn.BatchNorm2d / nn.ReLU modules)."""
import contextlib
import os


@contextlib.contextmanager
def forced_kernels():
    """Winograd 3x3 (wino._MIN_WG), bf16x6 1x1 forward / data gradient (gemm_bf6._MIN_ITEMS) and weight gradient (_MIN_WGRAD_ROWS),
    fp32-MFMA 1x1 + statistics (bn_act._C1 min_rows): every ELIGIBLE layer takes the hand-written kernel"""
    from distill_bev_amd import bn_act, gemm_bf6, wino
    old = (wino._MIN_WG, gemm_bf6._MIN_ITEMS, gemm_bf6._MIN_WGRAD_ROWS, bn_act._C1["min_rows"])
    wino._MIN_WG, gemm_bf6._MIN_ITEMS, gemm_bf6._MIN_WGRAD_ROWS, bn_act._C1["min_rows"] = 0, 1, 1, 1
    try:
        yield
    finally:
        wino._MIN_WG, gemm_bf6._MIN_ITEMS, gemm_bf6._MIN_WGRAD_ROWS, bn_act._C1["min_rows"] = old


@contextlib.contextmanager
def library_path():
    """DBEV_WINO=0 + DBEV_BF6=0 + bn_act.disabled() + no cancelled-bias / fp32-MFMA 1x1 path: convolutions on MIOpen, BatchNorm and
    ReLU as torch ops, bias gradients by ATen -- on the same (re-classed) module tree"""
    from distill_bev_amd import bn_act, colsum, gemm_bf6
    old_env = os.environ.get("DBEV_WINO")
    old = (gemm_bf6._ON, colsum._ON, bn_act._C1["enabled"])
    os.environ["DBEV_WINO"] = "0"
    gemm_bf6._ON, colsum._ON, bn_act._C1["enabled"] = False, False, False
    try:
        with bn_act.disabled():
            yield
    finally:
        gemm_bf6._ON, colsum._ON, bn_act._C1["enabled"] = old
        if old_env is None:
            os.environ.pop("DBEV_WINO", None)
        else:
            os.environ["DBEV_WINO"] = old_env


def kernels_ran(fn):
    """run fn() with the library's kernel event log on -> (fn's result, {kernel name: launches})"""
    from distill_bev_amd import _lib as L
    L.kernel_timing_read()
    L.kernel_timing(True)
    try:
        out = fn()
        log = L.kernel_timing_read()
    finally:
        L.kernel_timing(False)
    return out, {k: len(v) for k, v in log.items()}
