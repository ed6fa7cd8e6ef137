#!/bin/bash
# cold vs L2-warm duration of the forward stage kernels (GPU box): stage_variant 16384 issues every stage twice
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_warm
ZEGGS_OPTIONS="stage_variant=16384" rocprofv3 --kernel-trace -d $R/gpurun_out/prof_warm -o r -- python $R/tools/fwd_probe.py > $R/gpurun_out/prof_warm.log 2>&1
python $R/tools/rocpd_evenodd.py $(find $R/gpurun_out/prof_warm -name "*.db" | head -1)
rm -rf $R/gpurun_out/prof_warm
