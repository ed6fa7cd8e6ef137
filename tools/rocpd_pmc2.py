#!/usr/bin/env python
"""HBM traffic of the decoder step, forward and backward, and of the persistent decode kernel, from two rocprofv3 --pmc
passes (FETCH_SIZE, WRITE_SIZE; rocpd sqlite) on tools/fwdbwd_probe.py.  FETCH_SIZE is doubled (MI355X_MICROARCH.md:
gfx950 counts 128-B requests of wide coalesced streams as 64 B); WRITE_SIZE is uncalibrated.
usage: tools/rocpd_pmc2.py <fetch.db> <write.db> <out.json>"""
import json
import re
import sqlite3
import sys

T_TRAIN, T_DEC = 64, 601
BATCH, H, SP, ST, PI, PO = 32, 1024, 64, 64, 1134, 1131
XD = PI + SP + ST
W = 4 * (H * XD + 3 * H * (H + XD) + 3 * H * H + 3 * H * H + 3 * H * H + PO * H + H + 4 * 3 * H + PO)
ALGO = W + BATCH * 4 * (PI + SP + ST + 2 * H + PO + 2 * H)


def table(dbfile, counter):
    db = sqlite3.connect(dbfile)
    q = ("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by 1")
    return {k: (int(n), float(v)) for k, n, v in db.execute(q, (counter,))}


fetch, write = table(sys.argv[1], "FETCH_SIZE"), table(sys.argv[2], "WRITE_SIZE")


def group(tab, pred):
    n = sum(c for k, (c, v) in tab.items() if pred(k))
    kb = sum(v for k, (c, v) in tab.items() if pred(k))
    return n, kb


fam = lambda f: (lambda k: re.search(r"stage_k<\d+, %d," % f, k) is not None)  # noqa: E731
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/fwdbwd_probe.py (B=32, T=64 "
                 "forward + BPTT, 2 repetitions; B=1 decode of 600 frames, persistent kernel), gfx950, ROCm 7.2",
       "fetch_correction": "x2 (MI355X_MICROARCH.md)", "algorithmic_bytes_per_step": ALGO}
steps = 2 * (T_TRAIN - 1)
tpk = lambda k: "train_fwd_persistent_k" in k  # noqa: E731
bpk = lambda k: "train_bwd_persistent_k" in k  # noqa: E731
for name, f in (("forward", 0), ("backward", 1)):
    pred = fam(f)
    if f == 0 and group(fetch, tpk)[0]:       # the forward sweep is one persistent launch per rollout
        pred = tpk
        out["forward_kernel"] = "train_fwd_persistent_k (one launch per rollout; its weights are fetched once per rollout)"
    if f == 1 and group(fetch, bpk)[0]:       # ... and so is the BPTT sweep
        pred = bpk
        out["backward_kernel"] = "train_bwd_persistent_k (one launch per sweep; its weight tiles are fetched once per sweep)"
    nf, kf = group(fetch, pred)
    nw, kw = group(write, pred)
    fb, wb = 2 * 1024 * kf / steps, 1024 * kw / steps
    out[name] = {"launches_per_step": round(nf / steps, 3), "fetch_bytes_per_step": int(fb), "write_bytes_per_step": int(wb),
                 "traffic_bytes_per_step": int(fb + wb), "traffic_over_algorithmic": round((fb + wb) / ALGO, 3)}
out["traffic_bytes_per_step"] = out["forward"]["traffic_bytes_per_step"]
out["traffic_bytes_per_step_backward"] = out["backward"]["traffic_bytes_per_step"]
pk = lambda k: "decode_persistent_k" in k  # noqa: E731
nf, kf = group(fetch, pk)
nw, kw = group(write, pk)
if nf:
    out["decode_persistent"] = {"launches": nf, "frames": T_DEC - 1, "fetch_bytes_per_frame": int(2 * 1024 * kf / (T_DEC - 1)),
                                "write_bytes_per_frame": int(1024 * kw / (T_DEC - 1)), "weight_bytes": W,
                                "note": "the 75.7 MB of weights are fetched once per rollout, not per frame"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
