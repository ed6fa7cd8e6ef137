#!/bin/bash
# bias column sums inside the weight-gradient products (GemmArgs.asum): parity tests, then headline A/B (option gemm_asum)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O; : > $O/asum_ab.log
timeout 900 python -m pytest tests -m gpu -x -q -k "decoder or train_iteration or streams or distributed or giveup" 2>&1 | tail -2 | tee -a $O/asum_ab.log
for rep in 1 2 3; do
for m in 0 1; do
  v=$(ZEGGS_OPTIONS="gemm_direct=1,gemm_direct_shield=1,gemm_direct_depth=8,gemm_asum=$m" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "rep $rep [gemm_asum=$m] ms/frames: $v" | tee -a $O/asum_ab.log
done
done
