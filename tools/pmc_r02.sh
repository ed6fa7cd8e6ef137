#!/bin/bash
# round 2 (GPU box): kernel stats of the bench + HBM traffic counters of the decoder step (fwd, bwd) and the persistent decode
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/ks; rocprofv3 --kernel-trace --stats -d $O/ks -o k -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/ks.log 2>&1
DB=$(find $O/ks -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB $O/r02_bench_kernel_stats.csv
python $R/tools/rocpd_stages.py $DB $O/r02_stage_breakdown.csv
rm -rf $O/ks2; rocprofv3 --kernel-trace --stats -d $O/ks2 -o k -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras > $O/ks2.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/ks2 -name "*.db" | head -1) $O/r02_train_only_kernel_stats_12steps.csv
rm -rf $O/ks3; rocprofv3 --kernel-trace --stats -d $O/ks3 -o k -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/ks3.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/ks3 -name "*.db" | head -1) $O/r02_train_only_kernel_stats_6steps.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p -- python $R/tools/fwdbwd_probe.py > $O/pmc_$c.log 2>&1
done
python $R/tools/rocpd_pmc2.py $(find $O/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/pmc_WRITE_SIZE -name "*.db" | head -1) $O/r02_decoder_step_pmc.json
rm -rf $O/ks $O/ks2 $O/ks3 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
# one training iteration as a timeline (start offset, duration, queue of every dispatch) + how much of it ran concurrently
rm -rf $O/tl; rocprofv3 --kernel-trace -d $O/tl -o k -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/tl.log 2>&1
python $R/tools/rocpd_timeline.py $(find $O/tl -name "*.db" | head -1) $O/r02_iteration_timeline.csv 2> $O/r02_iteration_timeline_summary.txt
rm -rf $O/tl
