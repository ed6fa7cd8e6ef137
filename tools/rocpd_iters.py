#!/usr/bin/env python
"""Milestones of EVERY training iteration of a rocprofv3 kernel trace (rocpd sqlite), one line each: offsets (us, from the end of
the previous iteration's last radam_k) of the first dispatch, the style encoder's first convolution, the forward sweep, the loss
section, the BPTT sweep, its end and the iteration's end -- tools/rocpd_timeline.py shows one iteration in full, this one shows
whether that one is typical.  usage: tools/rocpd_iters.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("PRAGMA table_info(kernels)")]
pick = lambda *c: next((x for x in c if x in cols), None)  # noqa: E731
name, st, en = pick("name", "kernel_name"), pick("start", "start_timestamp"), pick("end", "end_timestamp")
rows = list(db.execute(f"select {name}, {st}, {en} from kernels order by {st}"))
rad = [i for i, r in enumerate(rows) if "radam_k" in r[0]]
rad = [i for k, i in enumerate(rad) if k + 1 == len(rad) or rows[rad[k + 1]][1] - rows[i][1] > 2_000_000]
print("iter  first  conv0  conv0_end  fwd_sweep  loss  bwd_sweep  bwd_end  end   (us)")
for n, (lo, hi) in enumerate(zip(rad[:-1], rad[1:])):
    t0 = rows[lo][2]
    it = rows[lo + 1: hi + 1]
    def at(sub, end=False, first=True):  # noqa: E306
        m = [r for r in it if sub in r[0]]
        if not m:
            return float("nan")
        r = m[0] if first else m[-1]
        return (r[2 if end else 1] - t0) / 1000.0
    print(f"{n + 1:4d} {(it[0][1] - t0) / 1000.0:6.0f} {at('gemm_streamk_kernel<256'):6.0f} {at('gemm_streamk_kernel<256', True):9.0f} "
          f"{at('train_fwd_persistent'):9.0f} {at('train_fwd_persistent', True):6.0f} {at('train_bwd_persistent'):9.0f} "
          f"{at('train_bwd_persistent', True):8.0f} {(it[-1][2] - t0) / 1000.0:6.0f}")
