#!/usr/bin/env python
"""Decoder forward rollout probe for counter collection: B=32, T=64, training-mode forward (gates saved for BPTT),
the same stage launches as inside bench.py's timed step.  usage: fwd_probe.py [stage_variant]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
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
ops.set_option("stage_variant", int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for _ in range(2):
    ops.decoder_core(*args)
torch.cuda.synchronize()
