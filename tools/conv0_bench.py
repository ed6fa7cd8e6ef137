#!/usr/bin/env python
"""The style encoder's first conv as the GEMM it runs as (M = B L = 12288, N = 512, K = 3 * 1134, bias + ReLU epilogue)."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
from zeggs import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N, K = 12288, 512, 3402
A = torch.randn(M * K, device=dev)
B = torch.randn(K * N, device=dev)
C = torch.zeros(M, N, device=dev)
bias = torch.randn(N, device=dev)
f = lambda: ops.gemm(A, B, C, M, N, K, (K, 1), (N, 1), (N, 1), bias=bias, act=1)  # noqa: E731
f()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    f()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"conv0 as GEMM M={M} N={N} K={K}: {dt * 1e6:.1f} us  {2.0 * M * N * K / dt / 1e12:.1f} TFLOP/s")
