#!/usr/bin/env python
"""Chained (run-ahead) stage launches vs plain launches: B=1 / B=2 decode, per-step time from the library's stage-sweep
events, hand-off error word, and the largest output difference."""
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
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1801
for B in (1, 2):
    args = bench.decode_args(de, dev, T)
    if B == 2:
        args = (args[0],) + tuple(a.repeat(2, *([1] * (a.dim() - 1))) if torch.is_tensor(a) and a.dim() >= 2 and a.shape[0] == 1 else a
                                  for a in args[1:])
    outs = {}
    for chain in (0, 1, 0, 1):
        ops.set_option("chain", chain)
        with torch.no_grad():
            ops.decoder_core(*args)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = ops.decoder_core(*args)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        err = ops.last_decoder_chain_errors()
        sweep = bench.sweep_ms(0) * 1e3 / (T - 1)
        outs[chain] = out
        print(f"B={B} chain={chain}: {sweep:.2f} us/step (stage sweep), wall {wall * 1e6 / (T - 1):.2f} us/step, "
              f"errors={err}, finite={bool(torch.isfinite(out[0]).all())}", flush=True)
    print(f"B={B}: max |chained - plain| = {max(float((a - b).abs().max()) for a, b in zip(outs[0], outs[1])):.3e}")
ops.set_option("chain", 0)
