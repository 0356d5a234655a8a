python tools/dbg_fold.py 2>&1 | tail -6
