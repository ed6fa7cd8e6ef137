#!/bin/bash
# GPU box: kernel statistics + one-iteration timeline of the FiLM decoder + GRU style encoder variant
# (tools/variants_probe.py, 3 timed iterations + 1 warm-up)
#   bash tools/variants_prof.sh      (writes gpurun_out/variants_kernel_stats.csv, variants_stage_launches.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
python $R/tools/variants_probe.py 5
rm -rf $O/vks; timeout 170 rocprofv3 --kernel-trace --stats -d $O/vks -o k -- python $R/tools/variants_probe.py 3 > $O/vks.log 2>&1
DB=$(find $O/vks -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB $O/variants_kernel_stats.csv
python $R/tools/rocpd_timeline.py $DB $O/variants_timeline.csv 12
python $R/tools/rocpd_bygrid.py $DB > $O/variants_stage_launches.txt; cat $O/variants_stage_launches.txt
head -12 $O/variants_kernel_stats.csv | cut -c1-160
rm -rf $O/vks
