#!/usr/bin/env python3
"""Summarise a rocprofv3 result (rocpd sqlite .db, or *_kernel_stats.csv) into the short
text table committed under profiles/.   usage: rocpd_summary.py <results.db|stats.csv> [top_n]"""
import csv
import sqlite3
import sys


def short(name, n=110):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def from_db(path, top):
    c = sqlite3.connect(path)
    rows = list(c.execute("select * from top_kernels"))
    out = ["%-112s %7s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for name, calls, total, avg, pct in rows[:top]:
        out.append("%-112s %7d %12.1f %12.3f %7.2f" % (short(name), calls, total, avg, pct))
    return "\n".join(out)


def from_csv(path, top):
    rows = list(csv.DictReader(open(path)))
    out = ["%-112s %7s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for r in rows[:top]:
        out.append("%-112s %7d %12.1f %12.3f %7.2f" % (
            short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3,
            float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return "\n".join(out)


if __name__ == "__main__":
    p = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    print(from_db(p, top) if p.endswith(".db") else from_csv(p, top))
