#!/usr/bin/env python
"""fp32 MFMA GEMM (gemm.hip) on a few reference shapes, for tuning: the training step's weight-gradient shapes (TN, K = 8160)
and the square 4096^3 case the hardware guide quotes (untuned LDS-tiled kernel 122 TF, tuned 147 TF).
usage: [ZEGGS_LIB=<alt .so>] [ZEGGS_OPTIONS=...] python tools/gemm_probe.py [reps]"""
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
from zeggs import ops  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def bench(name, M, N, K, layout):
    A = torch.randn(M * K, device=dev)
    B = torch.randn(K * N, device=dev)
    C = torch.zeros(M, N, device=dev)
    sa, sb = {"TN": ((1, M), (N, 1)), "NN": ((K, 1), (N, 1)), "NT": ((K, 1), (1, K))}[layout]
    f = lambda: ops.gemm(A, B, C, M, N, K, sa, sb, (N, 1))  # noqa: E731
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # correctness of whatever build is loaded: a corner block against fp64
    Am = (A.view(K, M).t() if layout == "TN" else A.view(M, K))[:48].double()
    Bm = (B.view(N, K).t() if layout == "NT" else B.view(K, N))[:, -40:].double()
    err = float(((Am @ Bm) - C[:48, -40:].double()).abs().max() / (Am @ Bm).abs().max())
    assert err < 5e-6 or os.environ.get("GEMM_PROBE_NOCHECK"), (name, err)
    print(f"{name:22s} {layout} M={M} N={N} K={K}: {dt * 1e6:8.1f} us  {2.0 * M * N * K / dt / 1e12:6.1f} TFLOP/s", flush=True)


bench("square 4096", 4096, 4096, 4096, "NN")
bench("square 4096", 4096, 4096, 4096, "TN")
bench("dW_hh", 3072, 1024, 8160, "TN")
bench("dW_ih0", 3072, 2286, 8160, "TN")
bench("dW_l2", 1131, 1024, 8160, "TN")
bench("style conv0 dW", 3402, 512, 12288, "TN")
bench("style conv0 fwd", 12288, 512, 3402, "NN")
