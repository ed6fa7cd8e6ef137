#!/bin/bash
# round 5, final tree: the parts of tools/pmc_r05.sh whose subject changed late in the round (kernel statistics at two step counts
# and of the default bench, the single-rank RCCL line; the timeline comes from tools/r05_session.sh).  Counter passes: unchanged
# kernels, not repeated.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
T="timeout 200"
for n in 12 6; do
  rm -rf $O/ks; $T rocprofv3 --kernel-trace --stats -d $O/ks -o k -- python $R/bench.py --steps $n --warmup 2 --no-cpu-baseline --no-extras > $O/ks.log 2>&1
  python $R/tools/rocpd_stats.py $(find $O/ks -name "*.db" | head -1) $O/r05_train_only_kernel_stats_${n}steps.csv
done
rm -rf $O/ks; $T rocprofv3 --kernel-trace --stats -d $O/ks -o k -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-generate > $O/ks.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/ks -name "*.db" | head -1) $O/r05_bench_kernel_stats.csv
rm -rf $O/ks
cd $R && $T python bench.py --force-process-group --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $O/r05_bench_rccl_single_rank.json 2> $O/rccl.err; tail -c 300 $O/rccl.err
python - <<PY
import json
d = json.loads([l for l in open("$O/r05_bench_rccl_single_rank.json").read().splitlines() if l.startswith("{")][-1])   # (RCCL prints its banner first)
json.dump(d, open("$O/r05_bench_rccl_single_rank.json", "w"), indent=1)
print("rccl single rank:", d["ms_per_step"], d["value"])
PY
head -8 $O/r05_train_only_kernel_stats_12steps.csv
