#!/bin/bash
# GPU box: kernel trace of bench.generate_30min (generate_gesture() on 30 minutes, streaming writer): what runs between two
# chunk launches of decode_persistent_k, and for how long the device is idle there
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/gp; timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d $O/gp -o k -- python -c "
import sys; sys.path[:0]=['$R','$R/ubisoft-laforge-zeroeggs_amd']
import torch, bench
print(bench.generate_30min(torch.device('cuda:0'))['total_s'])
" > $O/gp.log 2>&1
tail -2 $O/gp.log
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$O/gp/**/*.db", recursive=True)[0])
cols = [r[1] for r in db.execute("PRAGMA table_info(kernels)")]
pick = lambda *c: next((x for x in c if x in cols), None)
name, st, en = pick("name", "kernel_name"), pick("start", "start_timestamp"), pick("end", "end_timestamp")
rows = list(db.execute(f"select {name}, {st}, {en} from kernels order by {st}"))
idx = [i for i, r in enumerate(rows) if "decode_persistent_k" in r[0]]
idx = idx[-14:]
print("chunk launches:", len(idx))
for a, b in zip(idx[:-1], idx[1:]):
    gap = (rows[b][1] - rows[a][2]) / 1e3
    between = rows[a + 1:b]
    busy = sum((r[2] - r[1]) for r in between) / 1e3
    big = sorted(((r[2] - r[1]) / 1e3, r[0].replace("(anonymous namespace)::", "").split("(")[0][:40]) for r in between)[-3:]
    print(f"chunk {(rows[a][2]-rows[a][1])/1e6:7.2f} ms, then gap {gap:8.1f} us: {len(between)} kernels busy {busy:7.1f} us; longest {big}")
PY
rm -rf $O/gp
