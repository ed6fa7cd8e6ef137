"""Does one bench extra disturb the next?  nhidden_512_b32 before and after variants_film_gru_b32 in ONE process."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "ubisoft-laforge-zeroeggs_amd"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402
from zeggs import engine  # noqa: E402

dev = torch.device("cuda:0")
data = bench.build_dataset(n_train=8, n_unique=2)
ds = engine.DeviceDataset(data, bench.WINDOW, dev)
for name in sys.argv[1:] or ["nhidden", "variants", "nhidden"]:
    r = bench.nhidden_512_b32(ds, dev) if name == "nhidden" else bench.variants_b32(ds, dev)
    print(name, r["ms_per_step"], r.get("roofline", {}).get("us_per_step"), r.get("roofline", {}).get("backward_us_per_step"))
