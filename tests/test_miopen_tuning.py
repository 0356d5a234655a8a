"""Host-side checks of the shipped MIOpen solver tables (distill_bev_amd/miopen_db) and their activation logic; the numerics of
the kernels they select are checked on the GPU in test_gpu_miopen_tables.py."""
import glob
import importlib
import os


def _fresh(monkeypatch, **env):
    for k in ("MIOPEN_USER_DB_PATH", "DBEV_MIOPEN_DB", "PYTORCH_TUNABLEOP_ENABLED"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from distill_bev_amd import miopen_tuning as MT
    MT = importlib.reload(MT)
    return MT


def test_tables_are_text_records_for_gfx950_fp32_nhwc():
    from distill_bev_amd import miopen_tuning as MT
    fdb = glob.glob(os.path.join(MT._DB, "gfx950*.ufdb.txt"))
    pdb = glob.glob(os.path.join(MT._DB, "gfx950*.udb.txt"))
    assert len(fdb) == 1 and len(pdb) == 1
    lines = [l for l in open(fdb[0]).read().splitlines() if l]
    assert len(lines) >= 250
    for l in lines:
        key, _, val = l.partition("=")
        assert val and "FP32" in key and ("NHWC" in key or "NCHW" in key), l[:120]
        for rec in val.split(";"):
            solver, _, rest = rec.partition(":")
            assert solver.isidentifier() and float(rest.split(",")[0]) > 0.0        # solver : measured ms, workspace, algorithm
    # the step's heaviest problem is in the table: 256 -> 256, 3x3, 16 x 44 maps, 48 images, all three directions
    for d in ("F", "B", "W"):
        assert any(l.startswith("256-16-44-3x3-256-16-44-48-1x1-1x1-1x1-0-NHWC-NHWC-NHWC-FP32-" + d) for l in lines), d
    assert os.path.getsize(fdb[0]) + os.path.getsize(pdb[0]) < 512 * 1024
    gemm = os.path.join(MT._DB, "tunableop_gemm.csv")
    rows = open(gemm).read().splitlines()
    assert rows[0].startswith("Validator,") and any(r.startswith("GemmTunableOp_float") for r in rows)


def test_activation_respects_the_environment(monkeypatch):
    MT = _fresh(monkeypatch)
    path = MT.use_shipped_db()
    assert path and os.environ["MIOPEN_USER_DB_PATH"] == path and path != MT._DB
    assert sorted(os.path.basename(f) for f in glob.glob(os.path.join(path, "*"))) == \
        sorted(os.path.basename(f) for f in glob.glob(os.path.join(MT._DB, "*db.txt")))
    assert MT.use_shipped_db() == path                      # idempotent, one private copy per process
    MT = _fresh(monkeypatch, MIOPEN_USER_DB_PATH="/some/user/dir")
    assert MT.use_shipped_db() is None and os.environ["MIOPEN_USER_DB_PATH"] == "/some/user/dir"
    MT = _fresh(monkeypatch, DBEV_MIOPEN_DB="0")
    assert MT.use_shipped_db() is None and "MIOPEN_USER_DB_PATH" not in os.environ
    assert MT.use_shipped_gemm_table() is None              # no GPU here / switched off


def test_stale_tables_are_detected_and_reported_once(monkeypatch, recwarn):
    """tables keyed to another MIOpen build are silently ignored by the library: tables_status() says so (bench.py prints it)"""
    import warnings
    MT = _fresh(monkeypatch)
    assert MT.tables_status() == "off"                       # not activated yet
    path = MT.use_shipped_db()
    assert len(MT.shipped_keys()) == 1 and MT.shipped_keys()[0].startswith("gfx950")
    import torch
    real = torch.backends.cudnn.version
    try:
        torch.backends.cudnn.version = lambda: 3005000       # the build the tables were tuned under (3.5.0)
        assert MT.tables_status() == "active"
        # the library found nothing under its own key, searched, and recorded under ANOTHER key: stale
        open(os.path.join(path, "gfx950100.HIP.9_9_9_deadbeef.ufdb.txt"), "w").write("x=y\n")
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert MT.tables_status() == "stale" and MT.tables_status() == "stale"
        assert len([x for x in w if "solver tables" in str(x.message)]) == 1
        os.remove(os.path.join(path, "gfx950100.HIP.9_9_9_deadbeef.ufdb.txt"))
        MT._state["warned"] = False
        torch.backends.cudnn.version = lambda: 3006001       # another library version: stale before the first convolution
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert MT.tables_status() == "stale"
        assert any("3.6.1" in str(x.message) for x in w)
    finally:
        torch.backends.cudnn.version = real
