#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel calls / total / average.
usage: tools/rocpd_stats.py <results.db> [out.csv]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
out.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
for n, c, t, a, p in rows:
    out.writerow([n[:110], c, round(t, 1), round(a, 3), round(p, 3)])
