#!/bin/bash
# per-stage kernel durations (rocprofv3 kernel trace grouped by kernel + grid) for a list of ZEGGS_OPTIONS settings
# usage (on the GPU box): bash tools/prof_stages.sh tag1 "opts1" tag2 "opts2" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
while [ $# -ge 2 ]; do
  tag=$1; opts=$2; shift 2
  rm -rf $R/gpurun_out/prof_$tag
  ZEGGS_OPTIONS="$opts" rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o r -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$tag.log 2>&1
  DB=$(find $R/gpurun_out/prof_$tag -name "*.db" | head -1)
  python $R/tools/rocpd_stages.py $DB $R/gpurun_out/stages_$tag.csv
  python $R/tools/rocpd_stats.py $DB $R/gpurun_out/kernels_$tag.csv
  echo "== $tag ($opts)"; grep stage_k $R/gpurun_out/stages_$tag.csv | cut -d, -f1-3 --complement | head -0
  grep stage_k $R/gpurun_out/stages_$tag.csv | sed 's/.*stage_k/stage_k/' | cut -c1-80
  rm -rf $R/gpurun_out/prof_$tag
done
