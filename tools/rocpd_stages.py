#!/usr/bin/env python
"""Per-stage view of a rocprofv3 kernel trace (rocpd sqlite): dispatches grouped by (kernel, grid size), so the
forward/backward stage launches of stage_k (same kernel, different grids) are separated.
usage: tools/rocpd_stages.py <results.db> [out.csv]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("PRAGMA table_info(kernels)")]
if "--schema" in sys.argv:
    print(cols)
pick = lambda *c: next(x for x in c if x in cols)  # noqa: E731
name, gx, wx = pick("name", "kernel_name"), pick("grid_x", "grid_size_x", "grid_size"), pick("workgroup_x", "workgroup_size_x", "workgroup_size")
st, en = pick("start", "start_timestamp"), pick("end", "end_timestamp")
q = (f"select {name}, {gx}, {wx}, count(*), avg({en}-{st})/1000.0, min({en}-{st})/1000.0, sum({en}-{st})/1000.0 "
     f"from kernels group by {name}, {gx}, {wx} order by 7 desc")
out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else sys.stdout)
out.writerow(["kernel", "grid_x", "wg_x", "calls", "avg_us", "min_us", "total_us"])
for n, g, w, c, a, m, t in db.execute(q):
    out.writerow([n[:90], g, w, c, round(a, 3), round(m, 3), round(t, 1)])
