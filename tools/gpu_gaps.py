#!/usr/bin/env python3
"""GPU idle time between kernels from a rocprofv3 rocpd database (dev tool): sorts the dispatches by start, sums the gaps
(next start - latest end so far) by size class and lists the kernels that precede the largest ones.
usage: gpu_gaps.py <results.db> [last_n_ms]   (only the last `last_n_ms` of the trace: the steady-state steps)"""
import sqlite3
import sys

db = sys.argv[1]
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
if last_ms:
    t1 = rows[-1][2]
    rows = [r for r in rows if r[1] >= t1 - last_ms * 1e6]
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
print(f"{len(rows)} kernels over {span / 1e6:.1f} ms: busy {busy / 1e6:.1f} ms, idle {(span - busy) / 1e6:.1f} ms")
edges = [1e3, 2e3, 5e3, 1e4, 2e4, 5e4, 1e5, 1e6, 1e12]
cls = [[0, 0.0] for _ in edges]
big = []
end = rows[0][2]
short = [0, 0.0]
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    end = max(end, e0)
    g = s1 - end
    if e0 - s0 < 4e3:
        short[0] += 1; short[1] += e0 - s0
    if g > 0:
        for i, e in enumerate(edges):
            if g < e:
                cls[i][0] += 1; cls[i][1] += g
                break
        if g > 2e4:
            big.append((g, n0[:70], n1[:70]))
lo = 0
for (n, t), e in zip(cls, edges):
    if n:
        print(f"  gaps {lo / 1e3:6.0f} - {e / 1e3:9.0f} us: {n:6d}  total {t / 1e6:8.2f} ms")
    lo = e
print(f"kernels shorter than 4 us: {short[0]} ({short[1] / 1e6:.2f} ms busy)")
for g, a, b in sorted(big, reverse=True)[:25]:
    print(f"  {g / 1e3:8.1f} us after {a}  ->  {b}")
