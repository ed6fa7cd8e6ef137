#!/bin/bash
# the training rollout's prologue in five launches (option tp_prologue, default 1) against ten: bench + landmarks
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline"
one() { env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do
one ZEGGS_OPTIONS=tp_prologue=1
one ZEGGS_OPTIONS=tp_prologue=0
done
rm -rf $O/tl; rocprofv3 --kernel-trace -d $O/tl -o k -- python $R/bench.py --steps 16 --warmup 6 --no-cpu-baseline --no-extras > $O/tl.log 2>&1
DB=$(find $O/tl -name "*.db" | head -1)
python $R/tools/r06_windows.py $DB | head -9 | tail -3 | cut -c1-150
TL_PAIR=-8 python $R/tools/rocpd_timeline.py $DB $O/tl_steady.csv 2> $O/tl_steady.txt; cat $O/tl_steady.txt
rm -rf $O/tl
