#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc results (rocpd .db): mean counter value per kernel name.
usage: pmc_summary.py <results.db> [substring filter ...]"""
import sqlite3
import sys

db = sys.argv[1]
filt = sys.argv[2:]
c = sqlite3.connect(db)
rows = list(c.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events "
                      "group by name, counter_name order by avg(counter_value) desc"))
print("%-90s %-12s %6s %14s %12s" % ("kernel", "counter", "calls", "avg value(KB)", "avg dur(us)"))
for name, cn, n, v, d in rows:
    if filt and not any(f in name for f in filt):
        continue
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-90s %-12s %6d %14.1f %12.1f" % (name[:90], cn, n, v, d / 1e3))
