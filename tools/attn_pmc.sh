#!/bin/bash
# counter passes over the style encoder alone (tools/style_probe.py): where the attention kernels' wave cycles go
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf $O/ap$i
  timeout 170 rocprofv3 --pmc $P --kernel-trace -d $O/ap$i -o p -- python $R/tools/style_probe.py 3 > $O/ap$i.log 2>&1
done
python - <<PY
import sqlite3, glob, collections
for i in (1, 2, 3):
    f = glob.glob("$O/ap%d/**/*.db" % i, recursive=True)
    if not f: print("pass", i, "no db"); continue
    db = sqlite3.connect(f[0])
    rows = list(db.execute("select dispatch_id, kernel_name, counter_name, value, end - start from counters_collection "
                           "where kernel_name like '%attn_%' order by dispatch_id"))
    disp = collections.OrderedDict()
    for d, n, c, v, dur in rows:
        e = disp.setdefault(d, {"dur_us": dur / 1e3, "k": n.split("(")[0][-24:]})
        e[c] = e.get(c, 0.0) + v
    last = {}
    for e in disp.values(): last[e["k"]] = e
    for k, e in last.items():
        print("pass", i, k, " ".join(f"{c}={v:.4g}" for c, v in e.items() if c != "k"))
PY
rm -rf $O/ap1 $O/ap2 $O/ap3
