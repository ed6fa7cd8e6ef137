#!/bin/bash
# shader clock and matrix-pipe utilisation of the GEMM under test builds: GRBM_GUI_ACTIVE / 8 XCDs / duration, MFMA busy / SIMD cycles
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; Z=$R/ubisoft-laforge-zeroeggs_amd/zeggs
export GEMM_PROBE_NOCHECK=1
for v in "$@"; do
  rm -rf $O/gc; ZEGGS_LIB=$Z/libzeggs_$v.so timeout 100 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d $O/gc -o p -- python $R/tools/gemm_probe.py 1 > $O/gc.log 2>&1
  python - <<PY
import sqlite3, glob, collections
f = glob.glob("$O/gc/**/*.db", recursive=True)
db = sqlite3.connect(f[0])
rows = list(db.execute("select dispatch_id, counter_name, value, end - start from counters_collection where kernel_name like '%gemm_%kernel%' order by dispatch_id"))
disp = collections.OrderedDict()
for d, c, v, dur in rows:
    e = disp.setdefault(d, {"dur": dur / 1e3}); e[c] = e.get(c, 0.0) + v
ds = list(disp.values()); per = max(1, len(ds) // 7)
for j, nm in enumerate(["sq NN", "sq TN", "dW_hh", "dW_ih0", "dW_l2", "conv0 dW", "conv0 fwd"]):
    e = ds[min(len(ds) - 1, (j + 1) * per - 1)]
    clk = e["GRBM_GUI_ACTIVE"] / 8 / e["dur"] / 1e3
    print(f"$v {nm:10s} {e['dur']:8.1f} us clock {clk:.3f} GHz mfma_util {e['SQ_VALU_MFMA_BUSY_CYCLES'] / (e['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f} parked {e['SQ_WAIT_ANY'] / e['SQ_WAVE_CYCLES']:.3f} issue {e['SQ_ACTIVE_INST_ANY'] / e['SQ_WAVE_CYCLES']:.3f}")
PY
done
rm -rf $O/gc
