cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 1 4; do
rm -rf $R/gpurun_out/prof_dec
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dec -o r -- python $R/tools/decode_bench.py $B 501 > $R/gpurun_out/prof_dec.log 2>&1
python $R/tools/rocpd_stages.py $(find $R/gpurun_out/prof_dec -name "*.db" | head -1) $R/gpurun_out/stages_dec$B.csv
echo "B=$B"; grep stage_k $R/gpurun_out/stages_dec$B.csv | sed 's/.*stage_k/stage_k/' | cut -c1-90
done
rm -rf $R/gpurun_out/prof_dec
