#!/usr/bin/env python
"""Who paces the persistent BPTT sweep: polling time per workgroup and phase (library built with -DZEGGS_BPSTAT, ZEGGS_LIB)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd"), str(ROOT / "tests")]
import bench  # noqa: E402
import helpers  # noqa: E402
from zeggs import ops as _ops_diag  # noqa: E402
_ops_diag._CHAIN_DIAGNOSTICS = True      # keep the last decoder workspace for the read-back below
from zeggs import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
B, T = 32, 128
_, de, _ = bench.build_nets(dev)
stats = synth.make_stats()
s = {k: v.to(dev) for k, v in helpers.stats_tensors().items()}
clips = [synth.make_clip(T, seed=300 + b, stats=stats) for b in range(B)]
tt = lambda k: torch.as_tensor(np.stack([c[k] for c in clips])).to(dev)  # noqa: E731
pose0 = helpers.pack_pose(tt("Y_root_vel"), tt("Y_root_vrt"), tt("Y_lpos"), tt("Y_ltxy"), tt("Y_lvel"), tt("Y_lvrt"))[:, 0].contiguous()
speech = (torch.randn(B, T, 64, device=dev) * 0.5).requires_grad_(True)
style = torch.randn(B, T, 64, device=dev) * 0.5
for _ in range(2):
    out = ops.decoder_core(de, pose0, tt("Y_root_pos")[:, 0].contiguous(), tt("Y_root_rot")[:, 0].contiguous(), tt("Y_gaze_pos"),
                           speech, style, s["in_mean"], s["in_std"], s["out_mean"], s["out_std"], synth.DT)
    (out[0].sum() + out[1].sum() + out[2].sum()).backward()
torch.cuda.synchronize()
d, training, ws = ops._LAST_DECODER_WS
buf = (C.c_ulonglong * (4 * 256 * 4))()
ops._check(ops.lib().zeggs_bp_waits(C.byref(d), ops._p(ws), C.c_size_t(ws.numel()), buf), "waits")
raw = np.array(buf[:], dtype=np.float64) / 100.0 / (T - 1)       # us per step
both = raw[:2048].reshape(2, 256, 4)
w, e = both[0], both[1]
q = raw[2048:].reshape(256, 8)
# column (p + 1) & 3 of the wait for phase instance p: waits INTO P1 (p = 4s - 1 -> 0), P2 (-> 1), P3 (-> 2), P4 (-> 3)
for k, name in enumerate(["P1 (after P4 of the step before)", "P2", "P3", "P4"]):
    col = w[:, k]
    order = np.argsort(col)
    print(f"wait into {name}: mean {col.mean():.2f} us/step  min {col.min():.2f} (wg {order[0]})  p10 {np.percentile(col, 10):.2f}  "
          f"median {np.median(col):.2f}  max {col.max():.2f} (wg {order[-1]})   least-waiting wgs: {order[:6].tolist()}")
print(f"total polling per step: mean {w.sum(1).mean():.2f} us, workgroup 0: {w[0].sum():.2f}, min over workgroups {w.sum(1).min():.2f}")
for k in range(4):
    col = e[:, k]
    print(f"P{k + 1} products done -> arrived: mean {col.mean():.2f} us  min {col.min():.2f}  max {col.max():.2f} (wg {int(col.argmax())})  wg0 {col[0]:.2f}  wg1 {col[1]:.2f}  wg141 {col[141]:.2f}")
for wg in (0, 1, 2, 141, 200):
    print(f"P4 epilogue of wg {wg} (thread 0): reduce+sync {q[wg, 0]:.2f}  items+sync {q[wg, 1]:.2f}  stores issued {q[wg, 2]:.2f}  drained {q[wg, 3]:.2f}  barrier+flag {q[wg, 4]:.2f}")
