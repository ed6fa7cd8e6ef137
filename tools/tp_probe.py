#!/usr/bin/env python
"""Persistent training-forward rollout vs the stage-launch chain: forward sweep time per step (library events), output and
gradient agreement (the backward is the stage-kernel BPTT either way, fed by what the forward saved)."""
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
import os  # noqa: E402
if "TP_DUAL" in os.environ:      # batch 17..32: two 16-row chains in one launch (csrc/train_dual.hip) or the single-chain sweep
    ops.set_option("tp_dual", int(os.environ["TP_DUAL"]))
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 256)
_, de, _ = bench.build_nets(dev)
stats = synth.make_stats()
s = {k: v.to(dev) for k, v in helpers.stats_tensors().items()}
clips = [synth.make_clip(T, seed=300 + b, stats=stats) for b in range(B)]
tt = lambda k: torch.as_tensor(np.stack([c[k] for c in clips])).to(dev)  # noqa: E731
pose0 = helpers.pack_pose(tt("Y_root_vel"), tt("Y_root_vrt"), tt("Y_lpos"), tt("Y_ltxy"), tt("Y_lvel"), tt("Y_lvrt"))[:, 0].contiguous()
res = {}
for tp in (0, 1, 0, 1):
    ops.set_option("train_persistent", tp)
    torch.manual_seed(3)
    speech = (torch.randn(B, T, 64, device=dev) * 0.5).requires_grad_(True)
    style = (torch.randn(B, T, 64, device=dev) * 0.5).requires_grad_(True)
    de.zero_grad()
    for rep in range(2):
        pose, rp, rr = ops.decoder_core(de, pose0, tt("Y_root_pos")[:, 0].contiguous(), tt("Y_root_rot")[:, 0].contiguous(),
                                        tt("Y_gaze_pos"), speech, style, s["in_mean"], s["in_std"], s["out_mean"], s["out_std"],
                                        synth.DT)
    torch.cuda.synchronize()
    fwd = bench.sweep_ms(0) * 1e3 / (T - 1)
    torch.manual_seed(4)
    wp, wr, wq = torch.randn_like(pose), torch.randn_like(rp), torch.randn_like(rr)
    ((pose * wp).sum() + (rp * wr).sum() + (rr * wq).sum()).backward()
    torch.cuda.synchronize()
    res[tp] = (pose.detach(), rp.detach(), rr.detach(), {k: p.grad.clone() for k, p in de.named_parameters()}, speech.grad.clone())
    print(f"B={B} T={T} train_persistent={tp}: forward {fwd:.2f} us/step, finite={bool(torch.isfinite(pose).all())}", flush=True)
a, b = res[0], res[1]
print("max |out diff|:", [f"{float((x - y).abs().max()):.2e}" for x, y in zip(a[:3], b[:3])])
rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp_min(1e-12))  # noqa: E731
print("worst param-grad rel diff:", max(rel(b[3][k], a[3][k]) for k in a[3]), "dspeech:", rel(b[4], a[4]))
ops.set_option("train_persistent", 0)
