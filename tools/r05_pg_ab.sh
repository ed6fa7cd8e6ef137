#!/bin/bash
# single-rank RCCL path (bench.py --force-process-group): prepare_ahead x style_head_first
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O; : > $O/pg_ab.log
for rep in 1 2; do
for m in "0 0" "0 3" "1 3"; do
  set -- $m
  v=$(ZEGGS_PREPARE_AHEAD=$1 ZEGGS_STYLE_HEAD_FIRST=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --force-process-group 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('allreduce_exposed_ms'))")
  echo "rep $rep [process group, ahead=$1 head_first=$2] ms/frames/exposed: $v" | tee -a $O/pg_ab.log
done
done
