#!/bin/bash
# GPU_MAX_HW_QUEUES (bench.launch_env sets 8 for ranks and for the extras' child processes) x the engine's GEMM defaults
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O; : > $O/hwq_ab.log
for rep in 1 2; do
for q in "" 8; do
  for o in "" "gemm_direct=5"; do
    v=$(env ${q:+GPU_MAX_HW_QUEUES=$q} ZEGGS_OPTIONS="$o" timeout 300 python bench.py --extra v2_label_b64 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('ms_per_step'), d.get('value'))")
    echo "rep $rep v2_label_b64 [hw queues '$q', opts '$o'] ms/frames: $v" | tee -a $O/hwq_ab.log
    v=$(env ${q:+GPU_MAX_HW_QUEUES=$q} ZEGGS_OPTIONS="$o" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('ms_per_step'), d.get('value'))")
    echo "rep $rep headline     [hw queues '$q', opts '$o'] ms/frames: $v" | tee -a $O/hwq_ab.log
  done
done
done
