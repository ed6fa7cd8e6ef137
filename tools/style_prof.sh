#!/bin/bash
# isolated kernel durations of the style encoder's forward + backward (tools/style_probe.py) for a list of ZEGGS_OPTIONS settings
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
: > $O/style_prof.txt
for o in "$@"; do
  oo=$o; [ "$o" = "-" ] && oo=""
  rm -rf $O/sp
  ZEGGS_OPTIONS="$oo" timeout 170 rocprofv3 --kernel-trace --stats -d $O/sp -o k -- python $R/tools/style_probe.py 10 > $O/sp.log 2>&1
  echo "== ZEGGS_OPTIONS=[$o]  $(grep 'ms per pass' $O/sp.log)" >> $O/style_prof.txt
  python $R/tools/rocpd_stats.py $(find $O/sp -name "*.db" | head -1) $O/sp.csv > /dev/null 2>&1
  python - >> $O/style_prof.txt <<PY
import csv
rows = list(csv.DictReader(open("$O/sp.csv")))
for r in rows:
    k = r["kernel"].replace("(anonymous namespace)::", "").split("(")[0][:60]
    if float(r["avg_us"]) >= 8:
        print(f"  {k:60s} calls {r['calls']:>4} avg {float(r['avg_us']):8.1f} us")
PY
done
rm -rf $O/sp; cat $O/style_prof.txt
