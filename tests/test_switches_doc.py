"""docs/design/10_switches.md lists every DBEV_* environment switch the sources read (tools/list_switches.py regenerates it)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_switch_read_in_the_sources_is_in_the_table():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import list_switches
    doc = open(os.path.join(ROOT, "docs", "design", "10_switches.md")).read()
    missing = [k for k in list_switches.switches() if f"`{k}`" not in doc]
    assert not missing, f"regenerate docs/design/10_switches.md (python tools/list_switches.py > docs/design/10_switches.md): {missing}"
