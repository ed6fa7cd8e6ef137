#!/usr/bin/env python
"""How far the host runs ahead of the GPU in the training loop: wall time of eng.step() (enqueue only) per iteration, and the
same with a device synchronisation after every step (GPU time of a step that starts on an idle chip)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd"), str(ROOT / "tests")]
import bench  # noqa: E402
from zeggs import engine, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
ds = engine.DeviceDataset(bench.build_dataset(), bench.WINDOW, dev)
se, de, st = bench.build_nets(dev)
eng = engine.TrainEngine(se, de, st, ds, synth.PARENTS, synth.DT)
perm = np.random.default_rng(42).permutation(len(ds))
idx = lambda it: engine.shard_indices(perm, it % (len(ds) // bench.BATCH), bench.BATCH, 1, 0)  # noqa: E731
for it in range(5):
    eng.step(idx(it), bench.EXAMPLE_LEN)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for it in range(5, 25):
    a = time.perf_counter()
    eng.step(idx(it), bench.EXAMPLE_LEN)
    host.append((time.perf_counter() - a) * 1e3)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) * 1e3 / 20
t0 = time.perf_counter()
for it in range(25, 45):
    eng.step(idx(it), bench.EXAMPLE_LEN)
    if it + 1 < 45:
        eng.prefetch(idx(it + 1), bench.EXAMPLE_LEN)
torch.cuda.synchronize()
print(f"with prefetch: wall ms per step {(time.perf_counter() - t0) * 1e3 / 20:.2f}, batches picked up: {eng.prefetch_hits}")
print("host ms per step (enqueue only):", " ".join(f"{h:.1f}" for h in host))
print(f"wall ms per step: {tot:.2f}")
sy = []
for it in range(45, 55):
    a = time.perf_counter()
    eng.step(idx(it), bench.EXAMPLE_LEN)
    h = (time.perf_counter() - a) * 1e3
    torch.cuda.synchronize()
    sy.append(((time.perf_counter() - a) * 1e3, h))
print("synchronised after every step: total / enqueue ms:", " ".join(f"{t:.1f}/{h:.1f}" for t, h in sy))
import cProfile, pstats  # noqa: E401,E402
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for it in range(55, 60):
    eng.step(idx(it), bench.EXAMPLE_LEN)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
