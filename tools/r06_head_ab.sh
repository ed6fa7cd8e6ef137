#!/bin/bash
# the iteration's head after round 6's changes: landmarks of steady windows + untraced bench, against ZEGGS_EXAMPLE_IN_PLACE=0
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
run() {
  rm -rf $O/tl; env $2 rocprofv3 --kernel-trace -d $O/tl -o k -- python $R/bench.py --steps 16 --warmup 6 --no-cpu-baseline --no-extras $3 > $O/tl.log 2>&1
  echo "== $1"; python $R/tools/r06_windows.py $(find $O/tl -name "*.db" | head -1) | head -9 | tail -3 | cut -c1-150
}
B="python $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline"
one() { env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
run default A=1 ""
run "example not in place" ZEGGS_EXAMPLE_IN_PLACE=0 ""
for r in 1 2 3; do
one A=1
one ZEGGS_EXAMPLE_IN_PLACE=0
done
rm -rf $O/tl
