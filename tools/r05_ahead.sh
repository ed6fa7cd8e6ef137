#!/bin/bash
# A/B of TrainEngine.prepare_ahead (ZEGGS_PREPARE_AHEAD) x style_head_first on the headline bench, then the stream / give-up tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O; : > $O/ahead_ab.log
for rep in 1 2 3; do
for m in "0 3" "1 3" "1 2" "1 0"; do
  set -- $m
  v=$(ZEGGS_PREPARE_AHEAD=$1 ZEGGS_STYLE_HEAD_FIRST=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "rep $rep [ahead=$1 head_first=$2] ms/frames: $v" | tee -a $O/ahead_ab.log
done
done
timeout 900 python -m pytest tests -m gpu -x -q -k "streams or giveup or train_iteration or distributed" 2>&1 | tail -4 | tee -a $O/ahead_ab.log
