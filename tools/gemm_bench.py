#!/usr/bin/env python
"""Throughput of the generic strided fp32 MFMA GEMM on the shapes the training step uses."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
from zeggs import ops  # noqa: E402

dev = torch.device("cuda:0")


def bench(name, M, N, K, layout):
    A = torch.randn(M * K, device=dev)
    B = torch.randn(K * N, device=dev)
    C = torch.zeros(M, N, device=dev)
    if layout == "TN":      # dW = dy^T x : A(m,k) = dy[k][m], B(k,n) = x[k][n]
        sa, sb = (1, M), (N, 1)
    elif layout == "NN":
        sa, sb = (K, 1), (N, 1)
    else:                   # NT
        sa, sb = (K, 1), (1, K)
    f = lambda: ops.gemm(A, B, C, M, N, K, sa, sb, (N, 1))  # noqa: E731
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name:28s} {layout} M={M} N={N} K={K}: {dt * 1e6:8.1f} us  {2.0 * M * N * K / dt / 1e12:6.1f} TFLOP/s", flush=True)


import os
for tgt in [int(v) for v in os.environ.get("ZEGGS_GEMM_BENCH_TARGETS", "1536,2048,3072").split(",")]:
    ops.set_option("gemm_wg_target", tgt)
    print("wg target", tgt)
    bench("dW_hh (decoder)", 3072, 1024, 8160, "TN")
    bench("dW_ih0 (decoder)", 3072, 2286, 8160, "TN")
    bench("dW_l2 (decoder)", 1131, 1024, 8160, "TN")
    bench("dW_l0 (decoder)", 1024, 1262, 8160, "TN")
    bench("style conv0 dW", 3402, 512, 12288, "TN")
