#!/bin/bash
# per-iteration milestones with / without TrainEngine.prepare_ahead
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; : > $O/iters.log
for a in 0 1; do
  rm -rf $O/tl; ZEGGS_PREPARE_AHEAD=$a timeout 250 rocprofv3 --kernel-trace -d $O/tl -o k -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $O/tl.log 2>&1
  echo "== prepare_ahead=$a" | tee -a $O/iters.log
  python $R/tools/rocpd_iters.py $(find $O/tl -name "*.db" | head -1) | tee -a $O/iters.log
done
rm -rf $O/tl
