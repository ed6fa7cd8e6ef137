"""Does the priority of the stream the training step runs on matter?  The headline configuration (configs_v1 nets, B = 32) on the
default stream and on a high-priority stream (the weight-gradient stream is the library's lowest-priority one, the speech
encoder's a normal one).  usage: prio_probe.py [steps]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "ubisoft-laforge-zeroeggs_amd"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402
from zeggs import engine, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
data = bench.build_dataset(n_train=8, n_unique=2)
ds = engine.DeviceDataset(data, bench.WINDOW, dev)
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
for label, stream in (("default stream", None), ("high-priority stream", torch.cuda.Stream(priority=-1)), ("default stream", None)):
    se, de, st = bench.build_nets(dev)
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        eng = engine.TrainEngine(se, de, st, ds, synth.PARENTS, synth.DT)
        perm = np.random.default_rng(42).permutation(len(ds))
        for it in range(5):
            eng.step(engine.shard_indices(perm, it, bench.BATCH, 1, 0), bench.EXAMPLE_LEN)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(5, 5 + steps):
            eng.step(engine.shard_indices(perm, it, bench.BATCH, 1, 0), bench.EXAMPLE_LEN)
        torch.cuda.synchronize()
    print(f"{label:22s} {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per iteration")
