cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O; rm -rf $O/tl
rocprofv3 --kernel-trace -d $O/tl -o k -- python $R/bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-extras > $O/tl.log 2>&1
python $R/tools/rocpd_iterations.py $(find $O/tl -name "*.db" | head -1)
rm -rf $O/tl
