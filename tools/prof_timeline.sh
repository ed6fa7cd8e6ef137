#!/bin/bash
# ONE training iteration of bench.py as a timeline (rocprofv3 kernel trace -> tools/rocpd_timeline.py); extra arguments go to
# bench.py (e.g. --force-process-group, --no-wgrad-overlap).  Output: gpurun_out/timeline.csv + timeline_summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O
rm -rf $O/tl; rocprofv3 --kernel-trace -d $O/tl -o k -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras "$@" > $O/tl.log 2>&1
python $R/tools/rocpd_timeline.py $(find $O/tl -name "*.db" | head -1) $O/timeline.csv 2> $O/timeline_summary.txt
cat $O/timeline_summary.txt; tail -2 $O/tl.log | cut -c1-200
rm -rf $O/tl
