#!/usr/bin/env python
"""The two fold products of the decoder packs (N0 = W_ih0[:, pose] diag W2: 3072 x 1024, K = 1131 -> 1132; Mc: 1024 x 1024) as they
run today (NN, LDS-tiled stream-K) against the same products in TN form (A stored k-major: the scaled matrix written transposed) on
the direct kernel, and against plain tiles (ZEGGS_OPTIONS=gemm_streamk=0).  usage: python tools/fold_probe.py"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
from zeggs import ops  # noqa: E402

dev = torch.device("cuda:0")


def run(name, M, N, K, layout, reps=20):
    A = torch.randn(M * K, device=dev)
    B = torch.randn(K * N, device=dev)
    C = torch.zeros(M, N, device=dev)
    sa, sb = {"TN": ((1, M), (N, 1)), "NN": ((K, 1), (N, 1))}[layout]
    f = lambda: ops.gemm(A, B, C, M, N, K, sa, sb, (N, 1))  # noqa: E731
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    Am = (A.view(K, M).t() if layout == "TN" else A.view(M, K))[:48].double()
    err = float(((Am @ B.view(K, N)[:, -40:].double()) - C[:48, -40:].double()).abs().max())
    print(f"{name:10s} {layout} {M} x {N} x {K}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s   err {err:.1e}", flush=True)


for K in (1132, 1131):
    run("N0 fold", 3072, 1024, K, "NN")
    run("Mc fold", 1024, 1024, K, "NN")
run("N0 fold", 3072, 1024, 1132, "TN")
run("Mc fold", 1024, 1024, 1132, "TN")
