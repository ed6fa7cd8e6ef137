#!/bin/bash
# A/B of (ZEGGS_OPTIONS | bench flags) pairs on the headline bench: usage  bash tools/r05_ab2.sh tag "opts|flags" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=$1; shift
cd $R; mkdir -p $O; : > $O/${TAG}_ab.log
for rep in 1 2; do
for o in "$@"; do
  opts=${o%%|*}; flags=${o#*|}; [ "$opts" = "-" ] && opts=""
  v=$(ZEGGS_OPTIONS="$opts" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $flags 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline'].get('us_per_step'), d['roofline'].get('backward',{}).get('us_per_step'))")
  echo "rep $rep [$o] ms/frames/fwd_us/bwd_us: $v" | tee -a $O/${TAG}_ab.log
done
done
