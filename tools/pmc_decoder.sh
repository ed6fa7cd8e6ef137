#!/bin/bash
# HBM traffic of the decoder forward step (GPU box): two --pmc passes, kernel trace only (no other trace domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_$c -o p -- python $R/tools/fwd_probe.py > $R/gpurun_out/pmc_$c.log 2>&1
done
python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $R/gpurun_out/pmc_WRITE_SIZE -name "*.db" | head -1) $R/gpurun_out/decoder_step_pmc.json
