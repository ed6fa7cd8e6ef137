#!/bin/bash
# packs made ahead (behind the optimizer, queue 2 starts them BEFORE conv0) against packs released behind conv0's launch (ZEGGS_PREPARE_AHEAD=0)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline"
one() { env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do
one ZEGGS_PREPARE_AHEAD=1
one ZEGGS_PREPARE_AHEAD=0
done
for v in 0; do
rm -rf $O/tl; ZEGGS_PREPARE_AHEAD=$v rocprofv3 --kernel-trace -d $O/tl -o k -- python $R/bench.py --steps 16 --warmup 6 --no-cpu-baseline --no-extras > $O/tl.log 2>&1
echo "== ZEGGS_PREPARE_AHEAD=$v"; python $R/tools/r06_windows.py $(find $O/tl -name "*.db" | head -1) | head -10 | cut -c1-150
done
rm -rf $O/tl
