#!/bin/bash
# headline bench with the previous loss kernels (zeggs/libzeggs_lossold.so) vs the library as built, three repetitions
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O; : > $O/loss_ab2.log
Z=$R/ubisoft-laforge-zeroeggs_amd/zeggs
for rep in 1 2 3; do
for lib in libzeggs_lossold.so libzeggs_hip.so; do
  v=$(ZEGGS_LIB=$Z/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "rep $rep [$lib] ms/frames: $v" | tee -a $O/loss_ab2.log
done
done
