#!/usr/bin/env python
"""Every training iteration of a rocprofv3 kernel trace (rocpd sqlite), radam_k to radam_k: wall time, the two persistent sweeps,
the busy time of each queue outside the sweeps.  usage: tools/rocpd_iterations.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("PRAGMA table_info(kernels)")]
pick = lambda *c: next((x for x in c if x in cols), None)  # noqa: E731
name, st, en = pick("name", "kernel_name"), pick("start", "start_timestamp"), pick("end", "end_timestamp")
qid = pick("queue_id", "stream_id", "queue", "stream")
rows = list(db.execute(f"select {name}, {st}, {en}, {qid if qid else 0} from kernels order by {st}"))
rad = [i for i, r in enumerate(rows) if "radam_k" in r[0]]
for k, (a, b) in enumerate(zip(rad, rad[1:])):
    it = rows[a + 1: b + 1]
    t0 = rows[a][2]
    tot = (it[-1][2] - t0) / 1e3
    fwd = sum((e - s) / 1e3 for n, s, e, q in it if "train_fwd_persistent" in n)
    bwd = sum((e - s) / 1e3 for n, s, e, q in it if "train_bwd_persistent" in n)
    fs = next(((s - t0) / 1e3 for n, s, e, q in it if "train_fwd_persistent" in n), 0.0)
    fe = next(((e - t0) / 1e3 for n, s, e, q in it if "train_fwd_persistent" in n), 0.0)
    bs = next(((s - t0) / 1e3 for n, s, e, q in it if "train_bwd_persistent" in n), 0.0)
    be = next(((e - t0) / 1e3 for n, s, e, q in it if "train_bwd_persistent" in n), 0.0)
    print(f"iteration {k + 1:3d}: {tot:9.1f} us  {len(it):4d} dispatches | before the forward sweep {fs:7.1f}  sweep {fwd:7.1f}  "
          f"loss section {bs - fe:6.1f}  BPTT sweep {bwd:7.1f}  tail {tot - be:7.1f}")
