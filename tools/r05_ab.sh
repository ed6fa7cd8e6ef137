#!/bin/bash
# A/B of ZEGGS_OPTIONS settings on the headline bench (no extras): usage  bash tools/r05_ab.sh tag "opts1" "opts2" ...   ("-" = defaults)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=$1; shift
cd $R; mkdir -p $O; : > $O/${TAG}_ab.log
for rep in 1 2; do
for o in "$@"; do
  oo=$o; [ "$o" = "-" ] && oo=""
  v=$(ZEGGS_OPTIONS="$oo" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline'].get('us_per_step'), d['roofline'].get('backward',{}).get('us_per_step'))")
  echo "rep $rep [$o] ms/frames/fwd_us/bwd_us: $v" | tee -a $O/${TAG}_ab.log
done
done
