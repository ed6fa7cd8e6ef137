#!/bin/bash
# round 6 (GPU box): what profiles/r06_* of the final tree comes from (kernel statistics at two step counts and of the default bench,
# one iteration as a timeline, HBM traffic of the two sweeps, the single-rank RCCL line).  Counter passes are separate --pmc runs with
# --kernel-trace only.   bash tools/pmc_r06.sh   (writes gpurun_out/r06_*; what is to be judged is copied into profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
T="timeout 200"
for n in 12 6; do
  rm -rf $O/ks; $T rocprofv3 --kernel-trace --stats -d $O/ks -o k -- python $R/bench.py --steps $n --warmup 2 --no-cpu-baseline --no-extras > $O/ks.log 2>&1
  python $R/tools/rocpd_stats.py $(find $O/ks -name "*.db" | head -1) $O/r06_train_only_kernel_stats_${n}steps.csv
done
rm -rf $O/ks
# (second session: a STEADY window -- 22 iterations, so that the host is ahead of the GPU again under the tracer, and the window 8 from the
#  end: it contains the next batch's prefetch, which the last iteration of a run does not; landmarks of the steady windows beside it)
rm -rf $O/tl; $T rocprofv3 --kernel-trace -d $O/tl -o k -- python $R/bench.py --steps 16 --warmup 6 --no-cpu-baseline --no-extras > $O/tl.log 2>&1
TL_PAIR=-8 python $R/tools/rocpd_timeline.py $(find $O/tl -name "*.db" | head -1) $O/r06_iteration_timeline.csv 2> $O/r06_iteration_timeline_summary.txt
python $R/tools/r06_windows.py $(find $O/tl -name "*.db" | head -1) | head -10 | cut -c1-160 >> $O/r06_iteration_timeline_summary.txt
cat $O/r06_iteration_timeline_summary.txt; rm -rf $O/tl
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c; $T rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p -- python $R/tools/fwdbwd_probe.py > $O/pmc_$c.log 2>&1
done
python $R/tools/rocpd_pmc2.py $(find $O/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/pmc_WRITE_SIZE -name "*.db" | head -1) $O/r06_decoder_step_pmc.json | cut -c1-400
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cd $R && $T python bench.py --force-process-group --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $O/r06_bench_rccl_single_rank.json 2> $O/rccl.err; tail -c 300 $O/rccl.err
python - <<PY
import json
d = json.loads([l for l in open("$O/r06_bench_rccl_single_rank.json").read().splitlines() if l.startswith("{")][-1])   # (RCCL prints its banner first)
json.dump(d, open("$O/r06_bench_rccl_single_rank.json", "w"), indent=1)
print("rccl single rank:", d["ms_per_step"], d["value"])
PY
head -6 $O/r06_train_only_kernel_stats_12steps.csv
