#!/usr/bin/env python
"""Timeline of ONE training iteration from a rocprofv3 kernel trace (rocpd sqlite): every dispatch between the last two
radam_k launches with start offset, duration and queue, plus how much of the iteration the chip ran nothing / one / several
kernels at once.  usage: tools/rocpd_timeline.py <results.db> [out.csv] [cluster_gap_ms = 2]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("PRAGMA table_info(kernels)")]
pick = lambda *c: next((x for x in c if x in cols), None)  # noqa: E731
name, gx = pick("name", "kernel_name"), pick("grid_x", "grid_size_x", "grid_size")
st, en = pick("start", "start_timestamp"), pick("end", "end_timestamp")
qid = pick("queue_id", "stream_id", "queue", "stream")
rows = list(db.execute(f"select {name}, {gx}, {st}, {en}, {qid if qid else 0} from kernels order by {st}"))
GAP = int(float(sys.argv[3]) * 1e6) if len(sys.argv) > 3 else 2_000_000
rad = [i for i, r in enumerate(rows) if "radam_k" in r[0]]
# the optimizer step is applied in pieces (the decoder's slice early, on the weight-gradient stream): an iteration ends with the
# LAST radam_k of a cluster (no other one starts within the next 2 ms)
rad = [i for k, i in enumerate(rad) if k + 1 == len(rad) or rows[rad[k + 1]][2] - rows[i][2] > GAP]
if len(rad) < 2:
    sys.exit("need two radam_k dispatches")
# the shortest of the last few iterations (the bench's trailing iterations are separated by host-side event reads)
pairs = list(zip(rad[:-1], rad[1:]))[-5:]
lo, hi = min(pairs, key=lambda ab: rows[ab[1]][3] - rows[ab[0]][3])
import os
if os.environ.get("TL_PAIR"):       # TL_PAIR=-2: the second-to-last iteration (a MIDDLE one: the next batch's prefetch is in it), not the shortest
    lo, hi = list(zip(rad[:-1], rad[1:]))[int(os.environ["TL_PAIR"])]
it = rows[lo + 1: hi + 1]
t0 = rows[lo][3]
out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
out.writerow(["start_us", "dur_us", "queue", "grid_x", "kernel"])
for n, g, s, e, q in it:
    short = n.replace("(anonymous namespace)::", "").split("(")[0][:60]
    out.writerow([round((s - t0) / 1000.0, 2), round((e - s) / 1000.0, 2), q, g, short])
ev = sorted([(s, 1) for _, _, s, e, _ in it] + [(e, -1) for _, _, s, e, _ in it])
depth, last, acc = 0, t0, {0: 0.0, 1: 0.0, 2: 0.0}
for t, d in ev:
    acc[min(depth, 2)] += (t - last) / 1000.0
    last, depth = t, depth + d
tot = (it[-1][3] - t0) / 1000.0
print(f"iteration {tot:.1f} us, {len(it)} dispatches: idle {acc[0]:.1f} us, one kernel {acc[1]:.1f} us, two or more {acc[2]:.1f} us",
      file=sys.stderr)
