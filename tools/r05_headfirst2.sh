#!/bin/bash
# style_head_first on the single-rank RCCL path (bench.py --force-process-group), then the whole -m gpu suite with the default (3)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O; : > $O/headfirst_pg.log
for rep in 1 2; do
for m in 0 3; do
  v=$(ZEGGS_STYLE_HEAD_FIRST=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --force-process-group 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "rep $rep [process group, head_first=$m] ms/frames: $v" | tee -a $O/headfirst_pg.log
done
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee -a $O/headfirst_pg.log
