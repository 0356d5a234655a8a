#!/usr/bin/env python3
"""per-launch duration histogram of the kernels whose name contains a pattern, from a rocprofv3 rocpd .db (dev tool).
usage: kernel_hist.py <results.db> <pattern> [steps]"""
import sqlite3
import sys

db, pat = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
cols = [r[1] for r in c.execute(f"pragma table_info({kt[0]})")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
q = f"select {name_col}, (end - start) from {kt[0]} where {name_col} like ?"
rows = [d for _, d in c.execute(q, (f"%{pat}%",))]
rows.sort()
n = len(rows)
print(f"{pat}: {n} launches ({n / steps:.1f} per step), total {sum(rows) / 1e6 / steps:.3f} ms per step")
edges = [2e3, 5e3, 1e4, 2e4, 5e4, 1e5, 2e5, 5e5, 1e9]
lo = 0
for e in edges:
    sel = [d for d in rows if lo <= d < e]
    if sel:
        print(f"  {lo / 1e3:7.0f} - {e / 1e3:7.0f} us: {len(sel) / steps:7.1f} per step, {sum(sel) / 1e6 / steps:7.3f} ms per step")
    lo = e
