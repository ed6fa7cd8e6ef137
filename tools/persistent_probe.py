#!/usr/bin/env python
"""Persistent weight-stationary decode kernel vs the chain of stage launches (B=1): us per frame from the library's
sweep events and wall clock, largest output difference, error state."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
import bench  # noqa: E402
from zeggs import ops  # noqa: E402

dev = torch.device("cuda:0")
ops.set_option("timing", 1)
_, de, _ = bench.build_nets(dev)
for T in [int(x) for x in (sys.argv[1:] or ["1801", "301"])]:
    args = bench.decode_args(de, dev, T)
    outs = {}
    for pers in (0, 1, 0, 1):
        ops.set_option("persistent", pers)
        with torch.no_grad():
            ops.decoder_core(*args)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = ops.decoder_core(*args)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        sweep = bench.sweep_ms(0) * 1e3 / (T - 1)
        outs[pers] = out
        print(f"T={T} persistent={pers}: {sweep:.2f} us/frame (sweep), wall {wall * 1e6 / (T - 1):.2f} us/frame, "
              f"finite={bool(torch.isfinite(out[0]).all())}", flush=True)
    print(f"T={T}: max |persistent - stage| = " +
          ", ".join(f"{float((a - b).abs().max()):.3e}" for a, b in zip(outs[0], outs[1])))
ops.set_option("persistent", 1)
