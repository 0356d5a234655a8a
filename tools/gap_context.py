#!/usr/bin/env python3
"""kernels around the largest GPU idle gaps of a rocprofv3 rocpd database (dev tool).  usage: gap_context.py <db> [last_ms] [n]"""
import sqlite3, sys
db = sys.argv[1]; last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0; top = int(sys.argv[3]) if len(sys.argv) > 3 else 8
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
t1 = rows[-1][2]
rows = [r for r in rows if r[1] >= t1 - last_ms * 1e6]
t0 = rows[0][1]
gaps = []
end = rows[0][2]
for i in range(1, len(rows)):
    end = max(end, rows[i - 1][2])
    g = rows[i][1] - end
    if g > 1e5:
        gaps.append((g, i))
for g, i in sorted(gaps, key=lambda x: x[1])[:top * 3]:
    print(f"--- gap {g / 1e3:.0f} us at t = {(rows[i][1] - t0) / 1e6:.2f} ms")
    for j in range(max(0, i - 4), min(len(rows), i + 4)):
        n, s, e = rows[j]
        print(f"   {'>' if j == i else ' '} {(s - t0) / 1e6:9.3f} ms {(e - s) / 1e3:8.1f} us  {n[:110]}")
