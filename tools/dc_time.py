#!/usr/bin/env python
"""Slot stamps of the dual-chain training-forward rollout (library built with -DZEGGS_DCTIME, loaded via ZEGGS_LIB):
per wave of workgroups 0 and 255, step T - 2, the six slots A0 B0 A1 B1 A2 B2 with
[slot start, old products done, arrival seen, fresh products done, partial sums signalled, epilogue done (its owner wave only)]."""
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
ops.set_option("train_persistent", 1)
ops.set_option("tp_dual", 1)
speech = (torch.randn(B, T, 64, device=dev) * 0.5).requires_grad_(True)
style = torch.randn(B, T, 64, device=dev) * 0.5
for _ in range(2):
    out = ops.decoder_core(de, pose0, tt("Y_root_pos")[:, 0].contiguous(), tt("Y_root_rot")[:, 0].contiguous(), tt("Y_gaze_pos"),
                           speech, style, s["in_mean"], s["in_std"], s["out_mean"], s["out_std"], synth.DT)
torch.cuda.synchronize()
print("state", ops.lib().zeggs_persistent_state(1))
d, training, ws = ops._LAST_DECODER_WS
buf = (C.c_ulonglong * (2 * 8 * 6 * 6))()
ops._check(ops.lib().zeggs_tp_dual_stamps(C.byref(d), ops._p(ws), C.c_size_t(ws.numel()), buf), "stamps")
st = np.array(buf[:], dtype=np.uint64).reshape(2, 8, 6, 6).astype(np.float64) / 100.0      # us
names = ["A0", "B0", "A1", "B1", "A2", "B2"]
for wg in (0, 1):
    t0 = st[wg, :, 0, 0].min()
    print(f"workgroup {'0' if wg == 0 else '255'} (us since the first wave entered slot A0; start old| wait| fresh| sig [epi])")
    for w in range(8):
        row = []
        for sl in range(6):
            r = st[wg, w, sl] - t0
            own = (w == sl)
            row.append(f"{names[sl]} {r[0]:5.2f} {r[1] - r[0]:4.2f}|{r[2] - r[1]:4.2f}|{r[3] - r[2]:4.2f}|{r[4] - r[3]:4.2f}" +
                       (f" [{r[5] - r[4]:4.2f}]" if own else ""))
        print(f"  wave {w}: " + "   ".join(row))
    print(f"  step span (last signal of B2 - first start of A0): {st[wg, :, 5, 4].max() - t0:.2f} us")
