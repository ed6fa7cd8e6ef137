"""Profiling target: the FiLM decoder + GRU style encoder variant at the headline shape (bench.py: variants_film_gru_b32),
N steps, nothing else -- `rocprofv3 --kernel-trace --stats -- python tools/variants_probe.py [steps] [film|normal] [gru|attn]`."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "ubisoft-laforge-zeroeggs_amd"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402
from zeggs import engine, modules, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cond = sys.argv[2] if len(sys.argv) > 2 else "film"
sty = sys.argv[3] if len(sys.argv) > 3 else "gru"
dev = torch.device("cuda:0")
data = bench.build_dataset(n_train=8, n_unique=2)
ds = engine.DeviceDataset(data, bench.WINDOW, dev)
torch.manual_seed(1234)
se = modules.SpeechEncoder(synth.N_AUDIO, 64, bench.SP).to(dev).train()
de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, bench.SP, bench.ST, bench.H, 2, rnn_cond=cond).to(dev).train()
st = modules.StyleEncoder(synth.POSE_IN, 512, bench.ST, type=sty, use_vae=True).to(dev).train()
eng = engine.TrainEngine(se, de, st, ds, synth.PARENTS, synth.DT)
perm = np.random.default_rng(42).permutation(len(ds))
eng.step(engine.shard_indices(perm, 0, bench.BATCH, 1, 0), bench.EXAMPLE_LEN)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(1, 1 + steps):
    eng.step(engine.shard_indices(perm, it, bench.BATCH, 1, 0), bench.EXAMPLE_LEN)
th = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"{cond} + {sty}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms per iteration (host returned after {th / steps * 1e3:.2f} ms)")
import ctypes  # noqa: E402
from zeggs import ops  # noqa: E402
c, r = ctypes.c_long(0), ctypes.c_long(0)
ops.lib().zeggs_sweep_graph_stats(ctypes.byref(c), ctypes.byref(r))
print(f"sweep graphs: {c.value} captured, {r.value} replays")
