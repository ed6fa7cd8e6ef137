#!/usr/bin/env python
"""End-to-end sanity of the engine beyond per-iteration parity: N optimizer steps of the bench workload (B=32,
256-frame windows, synthetic clips), loss printed every 25 steps (it has to fall: the synthetic clips are smooth
band-limited signals the decoder can fit).  usage: train_curve.py [steps] [film]   (film: FiLM decoder + GRU style encoder,
the option surface on the stage kernels)"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
import bench  # noqa: E402
from zeggs import engine, modules, ops, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
data = bench.build_dataset(n_train=16)
ds = engine.DeviceDataset(data, bench.WINDOW, dev)
if len(sys.argv) > 2 and sys.argv[2] == "film":
    torch.manual_seed(1234)
    se = modules.SpeechEncoder(synth.N_AUDIO, 64, bench.SP).to(dev).train()
    de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, bench.SP, bench.ST, bench.H, 2, rnn_cond="film").to(dev).train()
    st = modules.StyleEncoder(synth.POSE_IN, 512, bench.ST, type="gru", use_vae=True).to(dev).train()
else:
    se, de, st = bench.build_nets(dev)
eng = engine.TrainEngine(se, de, st, ds, synth.PARENTS, synth.DT)
ops.manual_seed(1000)
torch.manual_seed(77)
perm = np.random.default_rng(42).permutation(len(ds))
losses = []
for it in range(steps):
    idx = engine.shard_indices(perm, it % (len(ds) // bench.BATCH), bench.BATCH, 1, 0)
    losses.append(float(eng.step(idx, bench.EXAMPLE_LEN)))
    if it % 25 == 0 or it == steps - 1:
        print(f"it {it:4d}  loss {losses[-1]:.4f}", flush=True)
first, last = np.mean(losses[:10]), np.mean(losses[-10:])
print(f"mean loss first 10 its {first:.3f} -> last 10 its {last:.3f}  ({'falls' if last < first - 2.0 else 'DOES NOT FALL'})")
assert np.isfinite(losses).all()
