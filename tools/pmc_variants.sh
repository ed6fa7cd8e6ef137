#!/bin/bash
# GPU box: HBM traffic of the FiLM decoder step on the stage kernels (FETCH_SIZE / WRITE_SIZE, separate --pmc passes with
# --kernel-trace only) -> gpurun_out/r06_film_step_pmc.json.  The style encoder is the attention one here, so that every stage_k
# launch of the trace is a decoder launch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/vpmc_$c; timeout 170 rocprofv3 --pmc $c --kernel-trace -d $O/vpmc_$c -o p -- python $R/tools/variants_probe.py 1 film attn > $O/vpmc_$c.log 2>&1
done
python - <<PY
import glob, json, re, sqlite3
def table(d, counter):
    db = sqlite3.connect(glob.glob(d + "/**/*.db", recursive=True)[0])
    q = "select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by 1"
    return {k: (int(n), float(v)) for k, n, v in db.execute(q, (counter,))}
fetch, write = table("$O/vpmc_FETCH_SIZE", "FETCH_SIZE"), table("$O/vpmc_WRITE_SIZE", "WRITE_SIZE")
B, H, SP, PI, PO = 32, 1024, 64, 1134, 1131
xd = PI + SP
W = 4 * (H * xd + 3 * H * (H + xd) + 3 * 3 * H * H + H * H + PO * H + 2 * H + 12 * H + PO)
ALGO = W + B * 4 * (xd + 2 * 2 * H + 2 * H + PO + 2 * H)
steps = 2 * 255      # warm-up iteration + one timed iteration
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on tools/variants_probe.py 1 film attn "
                 "(B=32, T=256, FiLM decoder on the stage kernels, 2 iterations), gfx950, ROCm 7.2", "fetch_correction": "x2 (MI355X_MICROARCH.md)",
       "algorithmic_bytes_per_step": ALGO}
for name, f in (("forward", 0), ("backward", 1)):
    pred = lambda k: re.search(r"stage_k<\d+, %d," % f, k) is not None
    nf = sum(c for k, (c, v) in fetch.items() if pred(k)); kf = sum(v for k, (c, v) in fetch.items() if pred(k))
    kw = sum(v for k, (c, v) in write.items() if pred(k))
    fb, wb = 2 * 1024 * kf / steps, 1024 * kw / steps
    out[name] = {"launches_per_step": round(nf / steps, 3), "fetch_bytes_per_step": int(fb), "write_bytes_per_step": int(wb),
                 "traffic_bytes_per_step": int(fb + wb), "traffic_over_algorithmic": round((fb + wb) / ALGO, 3)}
json.dump(out, open("$O/r06_film_step_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf $O/vpmc_FETCH_SIZE $O/vpmc_WRITE_SIZE
