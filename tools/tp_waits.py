#!/usr/bin/env python
"""Polling time per workgroup and phase of the persistent training-forward rollout (library built with -DZEGGS_TPSTAT)."""
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
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 128)
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
torch.cuda.synchronize()
d, training, ws = ops._LAST_DECODER_WS
buf = (C.c_ulonglong * (256 * 4))()
ops._check(ops.lib().zeggs_tp_waits(C.byref(d), ops._p(ws), C.c_size_t(ws.numel()), buf), "waits")
w = np.array(buf[:], dtype=np.float64).reshape(256, 4)[:, :3] / 100.0 / (T - 1)
# column (p + 1) % 3 of the wait for phase instance p = 3 (t - 1) + k - 1: into P1 (GRU layer 0), P2 (GRU layer 1), P3 (output stage)
for k, name in enumerate(["P1 GRU layer 0 (after the output stage of the step before)", "P2 GRU layer 1", "P3 output stage"]):
    col = w[:, k]
    print(f"wait into {name}: mean {col.mean():.2f} us/step  min {col.min():.2f} (wg {int(col.argmin())})  median {np.median(col):.2f}  "
          f"max {col.max():.2f} (wg {int(col.argmax())})")
print(f"total polling per step: mean {w.sum(1).mean():.2f} us, min over workgroups {w.sum(1).min():.2f}")
