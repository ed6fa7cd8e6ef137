#!/bin/bash
# GPU box: trace of tools/order_probe.py (two bench extras in one process) -- per-grid stage launch statistics and the gaps between
# consecutive stage launches of the LAST engine (is a slow follow-up run slow kernels, or launches that arrive late?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/ops; timeout 200 rocprofv3 --kernel-trace -d $O/ops -o k -- python $R/tools/order_probe.py "$@" > $O/ops.log 2>&1
tail -3 $O/ops.log
DB=$(find $O/ops -name "*.db" | head -1)
python $R/tools/rocpd_bygrid.py $DB
python - <<PY
import sqlite3
db = sqlite3.connect("$DB")
cols = [r[1] for r in db.execute("PRAGMA table_info(kernels)")]
pick = lambda *c: next((x for x in c if x in cols), None)
name, gx = pick("name", "kernel_name"), pick("grid_x", "grid_size_x", "grid_size")
st, en = pick("start", "start_timestamp"), pick("end", "end_timestamp")
rows = list(db.execute(f"select {name}, {gx}, {st}, {en} from kernels order by {st}"))
stg = [(g, s, e) for n, g, s, e in rows if "stage_k" in n]
last = stg[-3000:]
gaps = [(last[i + 1][1] - last[i][2]) / 1e3 for i in range(len(last) - 1)]
import statistics
print("last 3000 stage launches: median gap %.2f us, mean gap %.2f us, mean dur %.2f us" % (statistics.median(gaps), sum(gaps) / len(gaps), sum((e - s) for _, s, e in last) / len(last) / 1e3))
for i in range(1500, 1520):
    print(last[i][0], round((last[i][2] - last[i][1]) / 1e3, 2), "gap", round(gaps[i], 2))
PY
rm -rf $O/ops
