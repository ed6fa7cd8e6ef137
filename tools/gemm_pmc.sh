#!/bin/bash
# GEMM tuning run (GPU box): tools/gemm_probe.py on the shipped library and on every alternative build zeggs/libzeggs_v*.so, then
# counter passes on the shipped one (issue / wait split of the wave cycles, L2 hit rate, LDS conflicts, fabric reads).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; Z=$R/ubisoft-laforge-zeroeggs_amd/zeggs
mkdir -p $O
echo "== shipped" > $O/gemm_probe.log; timeout 60 python $R/tools/gemm_probe.py >> $O/gemm_probe.log 2>&1
for v in $Z/libzeggs_v*.so; do
  [ -f $v ] || continue
  echo "== $(basename $v)" >> $O/gemm_probe.log; ZEGGS_LIB=$v timeout 60 python $R/tools/gemm_probe.py >> $O/gemm_probe.log 2>&1
done
cat $O/gemm_probe.log
L=${GEMM_PMC_LIB:-$Z/libzeggs_hip.so}
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P3="FETCH_SIZE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf $O/gp$i
  ZEGGS_LIB=$L timeout 120 rocprofv3 --pmc $P --kernel-trace -d $O/gp$i -o p -- python $R/tools/gemm_probe.py 1 > $O/gp$i.log 2>&1
done
python - <<PY
import sqlite3, glob, collections
names = ["sq4096 NN", "sq4096 TN", "dW_hh", "dW_ih0", "dW_l2", "conv0 dW", "conv0 fwd"]
for i in (1, 2, 3):
    f = glob.glob("$O/gp%d/**/*.db" % i, recursive=True)
    if not f: print("pass", i, "no db"); continue
    db = sqlite3.connect(f[0])
    rows = list(db.execute("select dispatch_id, kernel_name, counter_name, value, end - start from counters_collection "
                           "where kernel_name like '%gemm_%kernel%' order by dispatch_id"))
    disp = collections.OrderedDict()
    for d, n, c, v, dur in rows:
        e = disp.setdefault(d, {"dur_us": dur / 1e3, "k": n.split("<")[0][-22:]})
        e[c] = e.get(c, 0.0) + v
    ds = list(disp.values())
    per = max(1, len(ds) // len(names))
    for j, nm in enumerate(names):
        g = ds[j * per:(j + 1) * per]
        if not g: continue
        e = g[-1]
        print(f"pass{i} {nm:10s} {e['dur_us']:8.1f} us " + " ".join(f"{k}={v:.4g}" for k, v in e.items() if k not in ("dur_us", "k")))
PY
rm -rf $O/gp1 $O/gp2 $O/gp3
