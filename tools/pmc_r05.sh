#!/bin/bash
# round 5 (GPU box): everything profiles/r05_* comes from.  Counter passes (--pmc) are separate runs with --kernel-trace only.
#   bash tools/pmc_r05.sh            (writes gpurun_out/r05_*; copy what is to be judged into profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
T="timeout 170"
# 1. kernel stats of the training step alone (12 and 6 timed steps: per-step figures = difference) and of the whole default bench
for n in 12 6; do
  rm -rf $O/ks; $T rocprofv3 --kernel-trace --stats -d $O/ks -o k -- python $R/bench.py --steps $n --warmup 2 --no-cpu-baseline --no-extras > $O/ks.log 2>&1
  python $R/tools/rocpd_stats.py $(find $O/ks -name "*.db" | head -1) $O/r05_train_only_kernel_stats_${n}steps.csv
done
rm -rf $O/ks; $T rocprofv3 --kernel-trace --stats -d $O/ks -o k -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-generate > $O/ks.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/ks -name "*.db" | head -1) $O/r05_bench_kernel_stats.csv
rm -rf $O/ks
# 2. one iteration as a timeline
rm -rf $O/tl; $T rocprofv3 --kernel-trace -d $O/tl -o k -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/tl.log 2>&1
python $R/tools/rocpd_timeline.py $(find $O/tl -name "*.db" | head -1) $O/r05_iteration_timeline.csv 2> $O/r05_iteration_timeline_summary.txt
cat $O/r05_iteration_timeline_summary.txt; rm -rf $O/tl
# 3. HBM traffic of the two sweeps and of the B=1 decode kernel (FETCH_SIZE x2 + WRITE_SIZE, separate passes)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c; $T rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p -- python $R/tools/fwdbwd_probe.py > $O/pmc_$c.log 2>&1
done
python $R/tools/rocpd_pmc2.py $(find $O/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/pmc_WRITE_SIZE -name "*.db" | head -1) $O/r05_decoder_step_pmc.json | cut -c1-400
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
# 4. MFMA utilisation of the weight-gradient GEMMs: the LDS-tiled stream-K kernel (gemm_direct=0) and the barrier-free direct one (default)
for mode in 0 1; do
rm -rf $O/pmc_gemm
ZEGGS_OPTIONS=gemm_direct=$mode ZEGGS_GEMM_BENCH_TARGETS=6144 $T rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_gemm -o p -- python $R/tools/gemm_bench.py > $O/pmc_gemm.log 2>&1
python - <<PY
import sqlite3, glob, json
db = sqlite3.connect(glob.glob("$O/pmc_gemm/**/*.db", recursive=True)[0])
rows = list(db.execute("select dispatch_id, kernel_name, grid_size_x, counter_name, value, end - start from counters_collection "
                       "where kernel_name like '%gemm_%kernel%'"))
disp = {}
for d, n, gx, c, v, dur in rows:
    e = disp.setdefault(d, {"dur_us": dur / 1e3, "kernel": n.split("(")[0][-40:]})
    e[c] = e.get(c, 0.0) + v
names = ["dW_hh 3072x1024 K=8160", "dW_ih0 3072x2286 K=8160", "dW_l2 1131x1024 K=8160", "dW_l0 1024x1262 K=8160", "style conv0 dW 3402x512 K=12288"]
ds = [e for _, e in sorted(disp.items()) if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE", 0) > 0]
out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on tools/gemm_bench.py (tools/pmc_r05.sh), gfx950, ZEGGS_OPTIONS=gemm_direct=$mode",
       "normalisation": "mfma_util = MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)", "kernels": {}}
per = len(ds) // len(names) if ds else 0
for i, nm in enumerate(names):
    grp = ds[i * per:(i + 1) * per]
    if grp:
        out["kernels"][nm] = {"launches": len(grp), "kernel": grp[-1]["kernel"],
                              "mfma_util": round(sum(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8 * 256 * 4) for e in grp) / len(grp), 4),
                              "avg_us_profiled": round(sum(e["dur_us"] for e in grp) / len(grp), 1)}
json.dump(out, open("$O/r05_gemm_mfma_util_direct$mode.json", "w"), indent=1)
print(json.dumps(out["kernels"]))
PY
grep TFLOP $O/pmc_gemm.log | head -6; rm -rf $O/pmc_gemm
done
# 5. who paces the two persistent sweeps (statistics build: -DZEGGS_BPSTAT -DZEGGS_TPSTAT, loaded through ZEGGS_LIB)
if [ -f $R/ubisoft-laforge-zeroeggs_amd/zeggs/libzeggs_stat.so ]; then
  ZEGGS_LIB=$R/ubisoft-laforge-zeroeggs_amd/zeggs/libzeggs_stat.so $T python $R/tools/bp_waits.py > $O/r05_bwd_persistent_waits.txt 2>&1
  ZEGGS_LIB=$R/ubisoft-laforge-zeroeggs_amd/zeggs/libzeggs_stat.so $T python $R/tools/tp_waits.py > $O/r05_train_persistent_waits.txt 2>&1
  tail -12 $O/r05_bwd_persistent_waits.txt; tail -5 $O/r05_train_persistent_waits.txt
fi
# 6. the data-parallel schedule through RCCL on ONE rank (the exchange of the final kernels, early decoder optimizer slices
#    behind their own all-reduce): same iteration as the headline line + the exposed exchange
cd $R && $T python bench.py --force-process-group --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $O/r05_bench_rccl_single_rank.json 2> $O/rccl.err; tail -c 300 $O/rccl.err
python - <<PY
import json
d = json.loads([l for l in open("$O/r05_bench_rccl_single_rank.json").read().splitlines() if l.startswith("{")][-1])   # (RCCL prints its banner first)
json.dump(d, open("$O/r05_bench_rccl_single_rank.json", "w"), indent=1)
print({k: d.get(k) for k in ("value", "ms_per_step", "allreduce_ms", "allreduce_exposed_ms", "allreduce_blocking_ms", "n1_equivalent_frames_per_sec")})
PY
