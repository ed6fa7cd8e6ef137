#!/bin/bash
# where do the ~70 us between the optimizer's last kernel and the next iteration's first kernel go?  Timelines of variants.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
run() {   # tag, env assignment, bench flags
  rm -rf $O/tl
  env $2 timeout 170 rocprofv3 --kernel-trace -d $O/tl -o k -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras $3 > $O/tl.log 2>&1
  python $R/tools/rocpd_timeline.py $(find $O/tl -name "*.db" | head -1) $O/gap_$1.csv 2> $O/gap_$1.txt
  echo "== $1: $(cat $O/gap_$1.txt)"; head -7 $O/gap_$1.csv | tail -6 | cut -c1-90
}
run default "A=1" ""
run noguard "ZEGGS_NO_GUARD=1" ""
run noprefetch "A=1" "--no-prefetch"
run nostepevents "ZEGGS_BENCH_NO_STEP_EVENTS=1" ""
rm -rf $O/tl
