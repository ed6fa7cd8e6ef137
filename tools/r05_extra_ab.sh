#!/bin/bash
# an extra of bench.py (own process) under several ZEGGS_OPTIONS settings: usage  bash tools/r05_extra_ab.sh <extra> "opts1" "opts2" ...  ("-" = defaults)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; X=$1; shift
cd $R; mkdir -p $O; : > $O/extra_${X}_ab.log
for rep in 1 2; do
for o in "$@"; do
  oo=$o; [ "$o" = "-" ] && oo=""
  v=$(ZEGGS_OPTIONS="$oo" timeout 300 python bench.py --extra $X 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('ms_per_step'), d.get('value'))")
  echo "rep $rep $X [$o] ms/frames: $v" | tee -a $O/extra_${X}_ab.log
done
done
