#!/usr/bin/env python3
"""Summarise the kernel dispatches of a rocprofv3 rocpd database (*_results.db) as a kernel_stats CSV.

rocprofv3 7.x writes an SQLite database by default; this reproduces the columns of its
`--stats --output-format csv` kernel summary so that profiles/ holds plain text.
usage: rocpd_kernel_stats.py results.db out.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, end - start from kernels").fetchall()
    agg = {}
    for name, d in rows:
        a = agg.setdefault(name, [0, 0, None, None])
        a[0] += 1
        a[1] += d
        a[2] = d if a[2] is None else min(a[2], d)
        a[3] = d if a[3] is None else max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([name, a[0], a[1], "%.1f" % (a[1] / a[0]), "%.2f" % (100.0 * a[1] / total), a[2], a[3]])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
