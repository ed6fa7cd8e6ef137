"""The weight-gradient GEMM shapes (dW = D^T X, K = B (T-1) = 8160) through the vendor BLAS that PyTorch-ROCm calls
(rocBLAS / hipBLASLt, fp32) and through this library's kernels, same timing loop: a reference point for gemm_streamk_kernel.
Round 4 (MI355X): vendor 116 / 121 / 122 / 136 / 136 TFLOP/s on dW_hh / dW_ih0 / dW_l2 / dW_l0 / conv0 dW and 151 on 4096^3;
own kernels 117 / 119 / 101 / 98 / 111 and 126.  Routing the plain products through rocblas_sgemm (dlopen, own handle +
256 MB workspace) was tried: equal on the two large shapes, SLOWER on the three small-output ones (83 / 93 / 95: the K-split
solutions PyTorch reaches are not what rocblas_sgemm picks) and the training iteration got 3 % slower (17.94 vs 17.38 ms) --
not adopted."""
import time

import torch

dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [("dW_hh 3072x1024 K=8160", 3072, 1024, 8160), ("dW_ih0 3072x2286 K=8160", 3072, 2286, 8160),
          ("dW_l2 1131x1024 K=8160", 1131, 1024, 8160), ("dW_l0 1024x1262 K=8160", 1024, 1262, 8160),
          ("style conv0 dW 3402x512 K=12288", 3402, 512, 12288), ("4096^3", 4096, 4096, 4096)]
for name, M, N, K in shapes:
    a = torch.randn(K, M, device=dev)
    b = torch.randn(K, N, device=dev)
    for _ in range(3):
        c = a.t() @ b
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        c = a.t() @ b
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:36s} {dt * 1e6:8.1f} us  {2.0 * M * N * K / dt / 1e12:6.1f} TFLOP/s")

# the same products through this library's kernels (zeggs_gemm: stream-K / split-K, gemm.hip)
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
from zeggs import ops  # noqa: E402
print("zeggs_gemm")
for name, M, N, K in shapes:
    a = torch.randn(K, M, device=dev)
    b = torch.randn(K, N, device=dev)
    c = torch.zeros(M, N, device=dev)
    f = lambda: ops.gemm(a, b, c, M, N, K, (1, M), (N, 1), (N, 1))  # noqa: E731
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:36s} {dt * 1e6:8.1f} us  {2.0 * M * N * K / dt / 1e12:6.1f} TFLOP/s")
