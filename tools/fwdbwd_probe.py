#!/usr/bin/env python
"""Counter-collection probe (round 2): B=32, T=64 training forward + BPTT through the stage kernels, then a B=1 decode of
600 frames through the persistent kernel."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
import bench  # noqa: E402
from zeggs import modules, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
B, T = 32, 64
torch.manual_seed(0)
de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, 64, 64, 1024, 2).to(dev)
s = {k: torch.as_tensor(v, dtype=torch.float32, device=dev) for k, v in synth.make_stats().items() if k.startswith("anim")}
args = (de, torch.randn(B, synth.POSE_OUT, device=dev), torch.zeros(B, 3, device=dev),
        torch.tensor([[1.0, 0, 0, 0]], device=dev).repeat(B, 1), torch.randn(B, T, 3, device=dev),
        (torch.randn(B, T, 64, device=dev) * 0.3).requires_grad_(True), torch.randn(B, T, 64, device=dev) * 0.3,
        s["anim_input_mean"], s["anim_input_std"], s["anim_output_mean"], s["anim_output_std"], synth.DT)
for _ in range(2):
    pose, rp, rr = ops.decoder_core(*args)
    (pose.sum() + rp.sum() + rr.sum()).backward()
torch.cuda.synchronize()
with torch.no_grad():
    ops.decoder_core(*bench.decode_args(de, dev, 601))
torch.cuda.synchronize()
