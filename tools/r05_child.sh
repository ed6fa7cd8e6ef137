#!/bin/bash
# why does the v2_label_b64 extra measure 7 % less inside the full bench than alone?  (a) the child's environment, (b) an idle parent context
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; : > $O/child_ab.log
run() { v=$(env "$@" timeout 300 python bench.py --extra v2_label_b64 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('ms_per_step'), d.get('value'))"); echo "[$*] $v" | tee -a $O/child_ab.log; }
run A=1
run OMP_NUM_THREADS=4 GPU_MAX_HW_QUEUES=8 HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "
import torch, time, sys
sys.path.insert(0, 'ubisoft-laforge-zeroeggs_amd')
x = torch.zeros(1 << 28, device='cuda'); s = [torch.cuda.Stream() for _ in range(3)]
from zeggs import ops; ops.side_stream('cuda:0'); torch.cuda.synchronize(); time.sleep(45)" &
PID=$!
sleep 8
run WITH_IDLE_PARENT=1
kill $PID 2>/dev/null; wait $PID 2>/dev/null
run A=2
