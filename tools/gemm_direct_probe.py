#!/usr/bin/env python
"""Round 5: the barrier-free / LDS-free stream-K TN product (gemm.hip: gemm_tn_direct_kernel) against the LDS-tiled stream-K
kernel on the weight-gradient shapes: full-matrix check against float64, then time per variant (option gemm_direct 0 / 1 / 2,
gemm_direct_wgs).  usage: python tools/gemm_direct_probe.py [reps]"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
from zeggs import ops  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
SHAPES = [("dW_hh", 3072, 1024, 8160, 3072, 1024), ("dW_ih0", 3072, 2286, 8160, 3072, 2288), ("dW_l2", 1131, 1024, 8160, 1132, 1024),
          ("dW_l0", 1024, 1262, 8160, 1024, 2288), ("conv0 dW", 3402, 512, 12288, 3402, 512), ("sq4096", 4096, 4096, 4096, 4096, 4096),
          ("odd", 197, 333, 1002, 200, 340)]


def run(name, M, N, K, lda, ldb, check):
    torch.manual_seed(1)
    A = torch.randn(K, lda, device=dev)          # A(m, k) = A[k][m]
    B = torch.randn(K, ldb, device=dev)
    C = torch.zeros(M, N, device=dev)
    f = lambda: ops.gemm(A, B, C, M, N, K, (1, lda), (ldb, 1), (N, 1))  # noqa: E731
    f()
    torch.cuda.synchronize()
    err = None
    if check:
        ref = A[:, :M].double().t() @ B[:, :N].double()
        err = float((C.double() - ref).abs().max() / ref.abs().max())
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt, err


for mode, wgs in ((0, 0), (1, 0), (2, 1), (2, 2), (3, 2), (3, 3)):
    ops.set_option("gemm_direct", mode)
    ops.set_option("gemm_direct_wgs", wgs)
    out = []
    for (name, M, N, K, lda, ldb) in SHAPES:
        dt, err = run(name, M, N, K, lda, ldb, check=True)
        assert err < 5e-6, (mode, wgs, name, err)
        out.append(f"{name} {dt * 1e6:7.1f}us {2.0 * M * N * K / dt / 1e12:5.1f}TF")
    print(f"gemm_direct={mode} wgs={wgs}: " + " | ".join(out), flush=True)
