#!/usr/bin/env python
"""stage_variant 16384 analysis: every forward stage launch is issued twice; average the 1st (cold) and 2nd (L2-warm)
launch of each pair separately.  usage: tools/rocpd_evenodd.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, grid_x, start, end from kernels where name like '%stage_k%' order by start"))
pairs = {}
i = 0
while i + 1 < len(rows):
    a, b = rows[i], rows[i + 1]
    if a[0] == b[0] and a[1] == b[1]:
        key = (a[0][-40:], a[1])
        p = pairs.setdefault(key, [0, 0.0, 0.0])
        p[0] += 1; p[1] += (a[3] - a[2]) / 1e3; p[2] += (b[3] - b[2]) / 1e3
        i += 2
    else:
        i += 1
for k, (n, c, w) in sorted(pairs.items(), key=lambda kv: -kv[1][0]):
    if n > 50:
        print(f"{k[0]} grid {k[1]:>7}: {n} pairs, first {c / n:6.2f} us, second (L2-warm weights) {w / n:6.2f} us")
