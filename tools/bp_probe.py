#!/usr/bin/env python
"""Persistent BPTT sweep vs the stage-launch sweep: backward sweep time per step (library events), agreement of every
parameter gradient and of dspeech / dstyle (same forward either way)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd"), str(ROOT / "tests")]
import bench  # noqa: E402
import helpers  # noqa: E402
from zeggs import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
ops.set_option("timing", 1)
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 256)
_, de, _ = bench.build_nets(dev)
stats = synth.make_stats()
s = {k: v.to(dev) for k, v in helpers.stats_tensors().items()}
clips = [synth.make_clip(T, seed=300 + b, stats=stats) for b in range(B)]
tt = lambda k: torch.as_tensor(np.stack([c[k] for c in clips])).to(dev)  # noqa: E731
pose0 = helpers.pack_pose(tt("Y_root_vel"), tt("Y_root_vrt"), tt("Y_lpos"), tt("Y_ltxy"), tt("Y_lvel"), tt("Y_lvrt"))[:, 0].contiguous()
res = {}
for bp in (0, 1, 0, 1):
    ops.set_option("bwd_persistent", bp)
    torch.manual_seed(3)
    speech = (torch.randn(B, T, 64, device=dev) * 0.5).requires_grad_(True)
    style = (torch.randn(B, T, 64, device=dev) * 0.5).requires_grad_(True)
    for rep in range(2):
        de.zero_grad()
        speech.grad = style.grad = None
        pose, rp, rr = ops.decoder_core(de, pose0, tt("Y_root_pos")[:, 0].contiguous(), tt("Y_root_rot")[:, 0].contiguous(),
                                        tt("Y_gaze_pos"), speech, style, s["in_mean"], s["in_std"], s["out_mean"], s["out_std"],
                                        synth.DT)
        torch.manual_seed(4)
        wp, wr, wq = torch.randn_like(pose), torch.randn_like(rp), torch.randn_like(rr)
        ((pose * wp).sum() + (rp * wr).sum() + (rr * wq).sum()).backward()
        torch.cuda.synchronize()
    bwd = bench.sweep_ms(1) * 1e3 / (T - 1)
    res[bp] = ({k: p.grad.clone() for k, p in de.named_parameters()}, speech.grad.clone(), style.grad.clone())
    print(f"B={B} T={T} bwd_persistent={bp} (state {ops.lib().zeggs_persistent_state(2)}): backward sweep {bwd:.2f} us/step, "
          f"finite={all(bool(torch.isfinite(g).all()) for g in res[bp][0].values())}", flush=True)
a, b = res[0], res[1]
rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp_min(1e-12))  # noqa: E731
worst = sorted(((rel(b[0][k], a[0][k]), k) for k in a[0]), reverse=True)[:4]
print("worst param-grad rel diffs:", [(f"{v:.2e}", k) for v, k in worst])
print("dspeech:", f"{rel(b[1], a[1]):.2e}", "dstyle:", f"{rel(b[2], a[2]):.2e}")
ops.set_option("bwd_persistent", 1)
