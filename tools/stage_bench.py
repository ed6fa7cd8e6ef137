#!/usr/bin/env python
"""Ablation micro-benchmark of the decoder stage kernels (GPU): times the B=32 forward+backward rollout
with the ablation switches of decoder_fast.hip (results are meaningless under ablation, only time matters)."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
from zeggs import modules, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, 64, 64, 1024, 2).to(dev)
s = {k: torch.as_tensor(v, dtype=torch.float32, device=dev) for k, v in synth.make_stats().items() if k.startswith("anim")}


def make(T):
    pose0 = torch.randn(B, synth.POSE_OUT, device=dev)
    rp, rr = torch.zeros(B, 3, device=dev), torch.tensor([[1.0, 0, 0, 0]], device=dev).repeat(B, 1)
    gaze = torch.randn(B, T, 3, device=dev) * 10
    speech = (torch.randn(B, T, 64, device=dev) * 0.3).requires_grad_(True)
    style = torch.randn(B, T, 64, device=dev) * 0.3
    return (de, pose0, rp, rr, gaze, speech, style, s["anim_input_mean"], s["anim_input_std"], s["anim_output_mean"],
            s["anim_output_std"], synth.DT)


def run(args, train):
    if train:
        p, a, b = ops.decoder_core(*args)
        (p.sum() + a.sum() + b.sum()).backward()
    else:
        with torch.no_grad():
            ops.decoder_core(*args)


def timed(args, train, reps=3):
    run(args, train)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run(args, train)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


A64, A192 = make(64), make(192)
NAMES = {0: "default", 4096 + 8192: "4+4 launches/step"}
for v, name in NAMES.items():
    ops.set_option("stage_variant", v)
    for train in (False, True):
        us = (timed(A192, train) - timed(A64, train)) / 128 * 1e6
        print(f"variant {v:2d} {name:16s} {'fwd+bwd' if train else 'fwd    '} {us:7.2f} us/step (marginal)", flush=True)
ops.set_option("stage_variant", 0)
