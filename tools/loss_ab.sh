#!/bin/bash
# loss kernels A/B: the library as built vs zeggs/libzeggs_lossold.so (the previous loss.hip), isolated call + kernel trace
# the comparison library (CPU container, from csrc/ after build.sh):  git show 1859741:ubisoft-laforge-zeroeggs_amd/csrc/loss.hip > /tmp/loss_old.hip
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -c /tmp/loss_old.hip -o /tmp/loss_old.o
#   hipcc --offload-arch=gfx950 -shared -fPIC -o ../zeggs/libzeggs_lossold.so $(ls build/*.o | grep -v /loss.o) /tmp/loss_old.o
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; : > $O/loss_ab.log
Z=$R/ubisoft-laforge-zeroeggs_amd/zeggs
for lib in $Z/libzeggs_lossold.so $Z/libzeggs_hip.so; do
  ZEGGS_LIB=$lib timeout 200 python $R/tools/loss_probe.py 2>&1 | tail -1 | tee -a $O/loss_ab.log
  ZEGGS_LIB=$lib timeout 200 python $R/tools/loss_probe.py 3 7 2>&1 | tail -1 | tee -a $O/loss_ab.log
  rm -rf $O/lp; ZEGGS_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats -d $O/lp -o k -- python $R/tools/loss_probe.py > /dev/null 2>&1
  python $R/tools/rocpd_stats.py $(find $O/lp -name "*.db" | head -1) 2>/dev/null | grep -i "loss_\|transpose" | tee -a $O/loss_ab.log
done
rm -rf $O/lp
cd $R; timeout 600 python -m pytest tests -m gpu -x -q -k "loss or train_iteration" 2>&1 | tail -3 | tee -a $O/loss_ab.log
