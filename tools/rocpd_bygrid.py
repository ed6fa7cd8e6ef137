#!/usr/bin/env python
"""Per (kernel, grid size) statistics from a rocprofv3 kernel trace (rocpd sqlite): tells the stage launches of one decoder
step apart (they share one kernel name).  usage: tools/rocpd_bygrid.py <results.db> [name-substring]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else "stage_k"
cols = [r[1] for r in db.execute("PRAGMA table_info(kernels)")]
pick = lambda *c: next((x for x in c if x in cols), None)  # noqa: E731
name, gx = pick("name", "kernel_name"), pick("grid_x", "grid_size_x", "grid_size")
st, en = pick("start", "start_timestamp"), pick("end", "end_timestamp")
acc = defaultdict(list)
for n, g, s, e in db.execute(f"select {name}, {gx}, {st}, {en} from kernels order by {st}"):
    if sub in n:
        acc[(n.replace("(anonymous namespace)::", "").split("(")[0][-40:], g)].append((e - s) / 1e3)
print(f"{'kernel':42s} {'grid_x':>9s} {'calls':>7s} {'avg_us':>8s} {'min_us':>8s} {'total_ms':>9s}")
for (n, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:42s} {g:9d} {len(v):7d} {sum(v) / len(v):8.2f} {min(v):8.2f} {sum(v) / 1e3:9.2f}")
