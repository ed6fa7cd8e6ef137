#!/usr/bin/env python
"""Phase stamps of chained launches (library built with -DZEGGS_CHTIME, loaded through ZEGGS_LIB): where the time of a
chained stage goes.  Prints, for the last launches of a B=1 rollout, microseconds relative to the launch's own start."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
import bench  # noqa: E402
from zeggs import ops as _ops_diag  # noqa: E402
_ops_diag._CHAIN_DIAGNOSTICS = True      # keep the last decoder workspace for the read-back below
from zeggs import ops  # noqa: E402

dev = torch.device("cuda:0")
_, de, _ = bench.build_nets(dev)
T = 120
args = bench.decode_args(de, dev, T)
ops.set_option("chain", 1)
with torch.no_grad():
    ops.decoder_core(*args)
    ops.decoder_core(*args)
torch.cuda.synchronize()
d, training, ws = ops._LAST_DECODER_WS
buf = (C.c_ulonglong * (16 * 2 * 16))()
ops._check(ops.lib().zeggs_decoder_chain_stamps(C.byref(d), training, ops._p(ws), C.c_size_t(ws.numel()), buf), "stamps")
st = np.array(buf[:], dtype=np.uint64).reshape(16, 2, 16).astype(np.float64) / 100.0      # us
K = 3 * (T - 1) + 1                     # launches in the rollout
last = K - 1
names = ["start", "lds-parked loaded", "all weights loaded", "flag seen+acquired", "x staged", "dots done", "reduced",
         "epilogue stores issued", "stores drained", "arrived"]
order = [(last - j) for j in range(11, -1, -1)]
t_ref = st[order[0] & 15, 0, 0]
for k in order:
    for wg in (0, 1):
        r = st[k & 15, wg]
        rel = r[:10] - r[0]
        print(f"launch {k} wg {'first' if wg == 0 else 'last '}: start@{r[0] - t_ref:8.2f}  " +
              "  ".join(f"{n.split()[0]}:{v:6.2f}" for n, v in zip(names[1:], rel[1:])))
    print()
