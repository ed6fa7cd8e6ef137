#!/bin/bash
# A/B of one library option in the headline iteration, one fresh process per run (stream -> hardware-queue mapping differs between
# processes, not inside one): tools/ab_option.sh "attn_bwd_one_launch=1" "attn_bwd_one_launch=0" [repeats]
R=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}; O=$R/gpurun_out; mkdir -p $O
A=$1; B=$2; N=${3:-3}
: > $O/ab_option.txt
for i in $(seq $N); do
  for o in "$A" "$B"; do
    ZEGGS_OPTIONS="$o" python $R/bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$o', d['ms_per_step'], d['value'], d.get('regions_ms'))" | tee -a $O/ab_option.txt
  done
done
