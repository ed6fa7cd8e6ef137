#!/bin/bash
# A/B of TrainEngine.style_head_first (ZEGGS_STYLE_HEAD_FIRST = 0..3) on the headline bench, then the engine parity tests with 3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O; : > $O/headfirst_ab.log
for rep in 1 2 3; do
for m in 0 1 2 3; do
  v=$(ZEGGS_STYLE_HEAD_FIRST=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "rep $rep [head_first=$m] ms/frames: $v" | tee -a $O/headfirst_ab.log
done
done
ZEGGS_STYLE_HEAD_FIRST=3 timeout 900 python -m pytest tests -m gpu -x -q -k "train_iteration or engine or giveup or style" 2>&1 | tail -5 | tee -a $O/headfirst_ab.log
