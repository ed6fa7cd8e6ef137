#!/bin/bash
# tools/cotenant_probe.py one setting per PROCESS (a second engine of a process maps its streams onto the hardware queues less
# luckily than the first: bench.py runs its extras in fresh processes for the same reason): usage cotenant_grid.sh "W:reserve ..."
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
: > $O/r06_reserve_grid.txt
for g in $1; do
  COMM_REUSE=${COMM_REUSE:-1} GRID=$g OUT=ct_one.txt timeout 300 python $R/tools/cotenant_probe.py ${2:-20} 2>/dev/null | grep -v "host time" | grep "stand-in\|no exchange" >> $O/r06_reserve_grid.txt
done
cat $O/r06_reserve_grid.txt | cut -c1-130
