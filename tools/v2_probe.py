"""configs_v2 shape (label conditioning, B = 64) with whatever ZEGGS_OPTIONS selects: bench.v2_label_b64 alone."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "ubisoft-laforge-zeroeggs_amd"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402
from zeggs import engine, ops  # noqa: E402

dev = torch.device("cuda:0")
data = bench.build_dataset(n_train=8, n_unique=2)
ds = engine.DeviceDataset(data, bench.WINDOW, dev)
ops.set_option("timing", 1)
r = bench.v2_label_b64(ds, dev)
print(r["value"], r["ms_per_step"], r["roofline"]["us_per_step"], r["roofline"]["backward_us_per_step"])
