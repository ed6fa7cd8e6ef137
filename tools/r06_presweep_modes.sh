#!/bin/bash
# the forward sweep starts 1.00-1.14 ms into the iteration depending on the window: the pre-sweep part of the fastest and of the slowest
# steady window of one trace side by side (queue, start, duration, kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
rm -rf $O/tl; rocprofv3 --kernel-trace -d $O/tl -o k -- python $R/bench.py --steps 16 --warmup 6 --no-cpu-baseline --no-extras "$@" > $O/tl.log 2>&1
DB=$(find $O/tl -name "*.db" | head -1)
python $R/tools/r06_windows.py $DB | head -10 | cut -c1-150
for k in -5 -6 -7 -8 -9 -10 -11 -12; do TL_PAIR=$k python $R/tools/rocpd_timeline.py $DB $O/tlw$k.csv 2> /dev/null; done
python - <<PY
import csv, glob
best = {}
for f in glob.glob("$O/tlw-*.csv"):
    rows = list(csv.reader(open(f)))[1:]
    fwd = [float(r[0]) for r in rows if "train_fwd" in r[4]]
    if fwd: best[f] = fwd[0]
lo, hi = min(best, key=best.get), max(best, key=best.get)
for tag, f in (("FAST", lo), ("SLOW", hi)):
    print("==", tag, best[f])
    for r in list(csv.reader(open(f)))[1:]:
        if float(r[0]) < best[f] + 1: print(f"q{r[2]} {float(r[0]):8.1f} {float(r[1]):7.1f}  {r[4][:50]}")
PY
rm -rf $O/tl $O/tlw-*.csv
