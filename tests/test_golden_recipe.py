"""tests/golden/make_golden.py, the recipe that pins the oracle to the reference: where /root/reference exists (the build container),
its multi-section invocation -- one interpreter per section since round 6 -- regenerates committed fixtures bit for bit.  (Round 5's
all-sections run died in the second detector section: stand-ins leaked between sections of one interpreter.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree only exists in the build container")
def test_two_sections_back_to_back_regenerate_bit_identically(tmp_path):
    env = dict(os.environ, DBEV_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(GOLD, "make_golden.py"), "fgmask", "center", "depth_map"], env=env, capture_output=True,
                       text=True, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2000:])
    made = sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    assert len(made) >= 3, made
    for f in made:
        a, b = np.load(os.path.join(tmp_path, f), allow_pickle=False), np.load(os.path.join(GOLD, f), allow_pickle=False)
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), (f, k)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree only exists in the build container")
def test_unknown_section_is_refused():
    r = subprocess.run([sys.executable, os.path.join(GOLD, "make_golden.py"), "no_such_section"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode != 0 and "unknown section" in (r.stderr + r.stdout)
