#!/bin/bash
# MFMA utilisation of the weight-gradient GEMMs (GPU box): SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x CUs x 4 SIMDs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_gemm
ZEGGS_GEMM_BENCH_TARGETS=6144 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_gemm -o p -- python $R/tools/gemm_bench.py > $R/gpurun_out/pmc_gemm.log 2>&1
python - <<PY
import sqlite3, glob, json
db = sqlite3.connect(glob.glob("$R/gpurun_out/pmc_gemm/**/*.db", recursive=True)[0])
rows = list(db.execute("select dispatch_id, kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name, value, end - start "
                       "from counters_collection where kernel_name like '%gemm_kernel%'"))
disp = {}
for d, n, gx, gy, gz, c, v, dur in rows:
    e = disp.setdefault(d, {"name": n[-60:], "grid": (gx, gy, gz), "dur_us": dur / 1e3})
    e[c] = e.get(c, 0.0) + v
agg = {}
for e in disp.values():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"] > 0:
        k = str(e["grid"])
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1; a[1] += e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8 * 256 * 4); a[2] += e["dur_us"]   # GUI_ACTIVE is summed over the 8 XCDs
out = {k: {"launches": n, "mfma_util": round(u / n, 4), "avg_us": round(t / n, 1)} for k, (n, u, t) in agg.items()}
print(json.dumps(out, indent=1))
json.dump(out, open("$R/gpurun_out/gemm_mfma_util.json", "w"), indent=1)
PY
grep TFLOP $R/gpurun_out/pmc_gemm.log | head -8
