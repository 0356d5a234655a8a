"""MIOpen solver tables tuned on MI355X for the convolutions of the shipped recipes.

ROCm 7.2's MIOpen carries no find database for gfx950 (share/miopen/db holds gfx942 and older): on a fresh machine every
convolution falls back to the immediate-mode heuristic plus a short hybrid search (~100 s at first touch), which picks
e.g. 16x64x32 tiles for the 58 x 100 maps of the BEVFormer recipe (45 TFLOP/s where the library's best kernel reaches 124).
`miopen_db/` holds the result of one exhaustive search (`torch.backends.cudnn.benchmark = True`, 9 + 7 minutes on one MI355X,
`tools/tune_miopen.sh`) over the shapes of the three convolution-bearing bench workloads at their shipped batch sizes: 281
problems -> ranked solvers (`*.ufdb.txt`) and tuned implicit-GEMM tile configurations (`*.udb.txt`), 190 KB of text.  With it the
library starts on its best kernels immediately: BEVDepth4D distillation step 157 -> 148 ms, BEVFormer step 351 -> 306 ms, first
step after 6-9 s instead of 100 s.  Shapes that are not in the table (another batch size, another image size) behave as before.

`tunableop_gemm.csv` is the same idea for the fp32 GEMMs of the BEVFormer recipe's 48 attention / FFN linears (40 000 x 256 queries):
PyTorch's TunableOp ranks the rocBLAS / hipBLASLt solutions per GEMM shape (67 s on one MI355X); `use_shipped_gemm_table()` loads the
69 results with tuning switched off (BEVFormer step 304 -> 293 ms).  Only the BEVFormer workload / recipe calls it.

`use_shipped_db()` points MIOPEN_USER_DB_PATH at a private copy (the library appends to the files; ranks must not share them)
and has to run before the process's first convolution.  A user-set MIOPEN_USER_DB_PATH wins; DBEV_MIOPEN_DB=0 disables it.
"""
import atexit
import glob
import os
import shutil
import tempfile

_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")
_state = {"path": None}


def use_shipped_db():
    """-> the directory MIOpen will use as its user database (None: left alone)."""
    if _state["path"] is not None:
        return _state["path"]
    if os.environ.get("DBEV_MIOPEN_DB", "1") == "0" or os.environ.get("MIOPEN_USER_DB_PATH"):
        return None
    files = glob.glob(os.path.join(_DB, "*db.txt"))
    if not files:
        return None
    dst = tempfile.mkdtemp(prefix="dbev_miopen_")
    for f in files:
        shutil.copy(f, dst)
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    _state["path"] = dst
    atexit.register(shutil.rmtree, dst, ignore_errors=True)
    return dst


def shipped_keys():
    """the MIOpen database keys (`<arch><cus>.HIP.<major>_<minor>_<patch>_<tweak>`) the shipped tables were tuned under"""
    return sorted({os.path.basename(f).rsplit(".", 2)[0] for f in glob.glob(os.path.join(_DB, "*db.txt"))})


def tables_status():
    """-> "off" | "active" | "stale".  The tables are plain text keyed by the MIOpen BUILD string in their file names; another
    ROCm / PyTorch build looks for other names and silently ignores them (the step is then ~7 % slower and the first one takes
    ~100 s of solver search).  Two checks, no 1 GB library scan: (a) before the first convolution the library's
    major.minor.patch (torch.backends.cudnn.version()) against the names; (b) afterwards, database files under another key in
    the private directory = the library searched and recorded for itself.  "stale" warns once."""
    dst = _state["path"]
    if dst is None:
        return "off"
    keys = shipped_keys()
    stale = None
    try:
        import torch
        v = int(torch.backends.cudnn.version() or 0)
        if v:
            mmp = f"{v // 1000000}_{v // 1000 % 1000}_{v % 1000}_"
            if not any(f".HIP.{mmp}" in k for k in keys):
                stale = f"running MIOpen {mmp[:-1].replace('_', '.')} vs tables {keys}"
    except Exception:                                     # torch without a MIOpen backend
        pass
    if stale is None:
        others = sorted({os.path.basename(f).rsplit(".", 2)[0] for f in glob.glob(os.path.join(dst, "*db.txt"))} - set(keys))
        if others:
            stale = f"the library recorded under {others}, tables are keyed {keys}"
    if stale and not _state.get("warned"):
        _state["warned"] = True
        import warnings
        warnings.warn("distill_bev_amd: the shipped MIOpen solver tables do not match this MIOpen build and are being ignored ("
                      + stale + "); re-tune with tools/tune_miopen.sh", RuntimeWarning)
    return "stale" if stale else "active"


def use_shipped_gemm_table():
    """Load the shipped TunableOp results (no tuning at run time).  -> the file prefix in use, or None."""
    import torch
    if os.environ.get("DBEV_MIOPEN_DB", "1") == "0" or os.environ.get("PYTORCH_TUNABLEOP_ENABLED") is not None:
        return None                                   # the environment decides
    src = os.path.join(_DB, "tunableop_gemm.csv")
    if not (os.path.exists(src) and torch.cuda.is_available() and hasattr(torch.cuda, "tunable")):
        return None
    dst = _state.get("gemm")
    if dst is None:
        d = tempfile.mkdtemp(prefix="dbev_gemm_")
        atexit.register(shutil.rmtree, d, ignore_errors=True)
        for i in range(torch.cuda.device_count()):     # TunableOp reads <name><device ordinal>.csv
            shutil.copy(src, os.path.join(d, f"tunableop_gemm{i}.csv"))
        dst = os.path.join(d, "tunableop_gemm.csv")
        _state["gemm"] = dst
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(False)
    torch.cuda.tunable.set_filename(dst, insert_device_ordinal=True)
    return dst
