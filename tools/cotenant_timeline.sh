#!/bin/bash
# Timeline of one iteration of tools/cotenant_probe.py (stand-in collective beside the tail): gpurun_out/cotenant_timeline.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O
rm -rf $O/tlc; GRID=${GRID:-32:32} OUT=ct_tmp.txt rocprofv3 --kernel-trace -d $O/tlc -o k -- python $R/tools/cotenant_probe.py 4 > $O/tlc.log 2>&1
python $R/tools/rocpd_timeline.py $(find $O/tlc -name "*.db" | head -1) $O/cotenant_timeline.csv 2> $O/cotenant_timeline_summary.txt
cat $O/cotenant_timeline_summary.txt; tail -2 $O/tlc.log | cut -c1-200
rm -rf $O/tlc
