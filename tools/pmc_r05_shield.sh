#!/bin/bash
# MFMA utilisation of the weight-gradient products through the SHIELD variant of the direct kernel (what TrainEngine runs):
# the pass of tools/pmc_r05.sh step 4 with ZEGGS_OPTIONS = gemm_direct=1,gemm_direct_shield=1,gemm_direct_depth=8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
OPTS="gemm_direct=1,gemm_direct_shield=1,gemm_direct_depth=8"
rm -rf $O/pmc_gemm
ZEGGS_OPTIONS=$OPTS ZEGGS_GEMM_BENCH_TARGETS=6144 timeout 170 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_gemm -o p -- python $R/tools/gemm_bench.py > $O/pmc_gemm.log 2>&1
python - <<PY
import sqlite3, glob, json
db = sqlite3.connect(glob.glob("$O/pmc_gemm/**/*.db", recursive=True)[0])
rows = list(db.execute("select dispatch_id, kernel_name, grid_size_x, counter_name, value, end - start from counters_collection "
                       "where kernel_name like '%gemm_%kernel%'"))
disp = {}
for d, n, gx, c, v, dur in rows:
    e = disp.setdefault(d, {"dur_us": dur / 1e3, "kernel": n.split("(")[0][-44:]})
    e[c] = e.get(c, 0.0) + v
names = ["dW_hh 3072x1024 K=8160", "dW_ih0 3072x2286 K=8160", "dW_l2 1131x1024 K=8160", "dW_l0 1024x1262 K=8160", "style conv0 dW 3402x512 K=12288"]
ds = [e for _, e in sorted(disp.items()) if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE", 0) > 0]
out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on tools/gemm_bench.py (tools/pmc_r05_shield.sh), gfx950, ZEGGS_OPTIONS=$OPTS",
       "normalisation": "mfma_util = MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)", "kernels": {}}
per = len(ds) // len(names) if ds else 0
for i, nm in enumerate(names):
    grp = ds[i * per:(i + 1) * per]
    if grp:
        out["kernels"][nm] = {"launches": len(grp), "kernel": grp[-1]["kernel"],
                              "mfma_util": round(sum(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8 * 256 * 4) for e in grp) / len(grp), 4),
                              "avg_us_profiled": round(sum(e["dur_us"] for e in grp) / len(grp), 1)}
json.dump(out, open("$O/r05_gemm_mfma_util_shield.json", "w"), indent=1)
print(json.dumps(out["kernels"]))
PY
grep TFLOP $O/pmc_gemm.log | head -6; rm -rf $O/pmc_gemm
