"""CPU: the C-ABI library loads and exports every symbol include/dbev_hip.h declares
(no compute calls -- there is no GPU here)."""
import os
import re

from conftest import ROOT


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "dbev_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dbev_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from distill_bev_amd import _lib
    h = _lib.lib()
    declared = _header_symbols()
    assert len(declared) >= 4
    for name in declared:
        assert hasattr(h, name), f"{name} declared in include/dbev_hip.h but not exported"
    # and the Python binding table covers exactly the header
    assert sorted(_lib.exported_symbols()) == declared


def test_identity_entry_points():
    from distill_bev_amd import _lib
    h = _lib.lib()
    assert h.dbev_abi_version() == 1
    assert h.dbev_target_arch() == b"gfx950"


def test_product_path_refuses_cpu_tensors():
    import pytest
    import torch
    from distill_bev_amd import _lib
    from distill_bev_amd.bev_pool import bev_pool
    with pytest.raises(_lib.DbevHipError):
        bev_pool(torch.zeros(4, 8), torch.zeros(4, 4, dtype=torch.long), 1, 1, 2, 2)


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "distill_bev_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
