#!/bin/bash
# A/B of one environment switch of the engine in the headline iteration, one fresh process per run:
#   tools/ab_env.sh ZEGGS_DEFER_STYLE_WGRADS 1 0 [repeats]
R=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}; O=$R/gpurun_out; mkdir -p $O
V=$1; A=$2; B=$3; N=${4:-3}
: > $O/ab_env.txt
for i in $(seq $N); do
  for o in "$A" "$B"; do
    env $V=$o python $R/bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$V=$o', d['ms_per_step'], d['value'])" | tee -a $O/ab_env.txt
  done
done
