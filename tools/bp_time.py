#!/usr/bin/env python
"""Phase stamps of the persistent BPTT sweep (library built with -DZEGGS_BPTIME, loaded through ZEGGS_LIB)."""
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
B, T = 32, 64
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
print("state", ops.lib().zeggs_persistent_state(2))
d, training, ws = ops._LAST_DECODER_WS
buf = (C.c_ulonglong * (3 * 2 * 32))()
ops._check(ops.lib().zeggs_bp_stamps(C.byref(d), ops._p(ws), C.c_size_t(ws.numel()), buf), "stamps")
st = np.array(buf[:], dtype=np.uint64).reshape(3, 2, 32).astype(np.float64) / 100.0
names = ["start", "P1 waited", "mma", "epilogue", "arrived", "P2 waited", "mma", "epilogue", "arrived", "P3 waited", "mma",
         "epilogue", "arrived", "P4 waited", "mma", "epilogue", "arrived"]
for k in range(3):
    for wg in (0, 1):
        r = st[k, wg, :17]
        print(f"step t={3 - k} wg {'0  ' if wg == 0 else '255'}: " + "  ".join(f"{n}:{v - r[0]:6.2f}" for n, v in zip(names[1:], r[1:])))
        if wg == 0 and st[k, 0, 17] > 0:
            print("      workgroup 0, P4 root path: reduced+synced %.2f  inputs loaded %.2f  applied %.2f" % tuple(st[k, 0, 17:20] - r[0]))
