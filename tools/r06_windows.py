#!/usr/bin/env python
"""Per iteration (radam_k to radam_k) of a rocprofv3 kernel trace: wall time, dispatches, and where the batch gather / the
first kernel of the caller's chain / conv0 / the two sweeps start.  usage: tools/r06_windows.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("PRAGMA table_info(kernels)")]
pick = lambda *c: next((x for x in c if x in cols), None)  # noqa: E731
name, st, en = pick("name", "kernel_name"), pick("start", "start_timestamp"), pick("end", "end_timestamp")
qid = pick("queue_id", "stream_id", "queue", "stream")
rows = list(db.execute(f"select {name}, {st}, {en}, {qid} from kernels order by {st}"))
rad = [i for i, r in enumerate(rows) if "radam_k" in r[0]]
rad = [i for k, i in enumerate(rad) if k + 1 == len(rad) or rows[rad[k + 1]][1] - rows[i][1] > 2_000_000]
for lo, hi in list(zip(rad[:-1], rad[1:]))[-14:]:
    it = rows[lo + 1: hi + 1]
    t0 = rows[lo][2]
    def first(sub, q=None):
        for n, s, e, qq in it:
            if sub in n and (q is None or qq == q):
                return f"{(s - t0) / 1e3:8.1f} q{qq}"
        return "       -   "
    def last(sub):
        r = [(e - t0) / 1e3 for n, s, e, qq in it if sub in n]
        return f"{r[-1]:8.1f}" if r else "    -"
    q1 = [(n, s, e) for n, s, e, qq in it if qq == it[-1][3] and "copyBuffer" not in n]
    f1 = f"first {q1[0][0].split('(')[0][-16:]:>16s} {(q1[0][1] - t0) / 1e3:6.1f}"
    print(f"{f1} | conv0 {first('gemm_streamk_kernel<256')} | " + f"{(it[-1][2] - t0) / 1e3:9.1f} us {len(it):4d} disp | gather {first('gather_windows')} .. {last('normalize_rows')} | pad_rows {first('pad_rows')} | "
          f"fwd {first('train_fwd')} | bwd {first('train_bwd')} | truth-transposes {sum('transpose_k' in n for n, *_ in it)}")
