#!/usr/bin/env python
"""Round 6 experiment: the fp32 TN products of the training tail on the bf16 matrix cores with an fp32-exact
three-plane operand split (csrc/gemm_split.hip, option "gemm_split_bf16" = 3 / 6 / 9 plane products, default 0 = off) against the
native fp32 MFMA kernel (the shielded direct kernel the training engine uses) on the five weight-gradient shapes: error of each
against a float64 product (max and RMS, relative to the RMS of the exact result) and time alone on the chip (incl. the zero fill
of C).  usage: python tools/gemm_split_probe.py [reps] -> gpurun_out/r06_gemm_split_bf16.txt"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
from zeggs import ops  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
# name, M, N, K, lda, ldb : C[M, N] = A[K, lda][:, :M]^T B[K, ldb][:, :N]   (dW = dy^T x over K = 255 x 32 frames)
SHAPES = [("dW_hh", 3072, 1024, 8160, 3072, 1024), ("dW_ih0", 3072, 2286, 8160, 3072, 2288), ("dW_l2", 1131, 1024, 8160, 1132, 1024),
          ("dW_l0", 1024, 1262, 8160, 1024, 2288), ("conv0 dW", 3402, 512, 12288, 3404, 512), ("sq4096", 4096, 4096, 4096, 4096, 4096)]
ops.set_option("gemm_direct", 1)
ops.set_option("gemm_direct_shield", 1)
ops.set_option("gemm_direct_depth", 8)
lines = [f"# tools/gemm_split_probe.py {reps}: C = A^T B on one MI355X, alone on the chip; native = gemm_tn_direct_shield_kernel (fp32 MFMA);"
         " split n = gemm_tn_split_kernel with n bf16 plane products; errors against a float64 product, relative to the RMS of the exact result;"
         " TFLOP/s are fp32-EQUIVALENT (2 M N K / time)"]
ok6 = ok9 = True
for (name, M, N, K, lda, ldb) in SHAPES:
    torch.manual_seed(1)
    A = torch.randn(K, lda, device=dev)
    B = torch.randn(K, ldb, device=dev)
    ref = A[:, :M].double().t() @ B[:, :N].double()
    scale = float(ref.pow(2).mean().sqrt())
    row = {}
    for np_ in (0, 9, 6, 3):
        ops.set_option("gemm_split_bf16", np_)
        C = torch.zeros(M, N, device=dev)
        f = lambda: ops.gemm(A, B, C, M, N, K, (1, lda), (ldb, 1), (N, 1))  # noqa: E731
        f()
        torch.cuda.synchronize()
        err = (C.double() - ref)
        emax, erms = float(err.abs().max()) / scale, float(err.pow(2).mean().sqrt()) / scale
        del err
        time.sleep(0.2)                      # (the float64 reference product above pulls the clocks down for a while)
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        e1.synchronize()
        dt = e0.elapsed_time(e1) * 1e-3 / reps
        row[np_] = (emax, erms, dt)
    n0 = row[0]
    s = f"{name:9s} {M} x {N} x {K}: native max {n0[0]:.2e} rms {n0[1]:.2e} {n0[2] * 1e6:7.1f} us {2.0 * M * N * K / n0[2] / 1e12:6.1f} TF"
    for np_ in (9, 6, 3):
        e = row[np_]
        s += f" | split {np_}: max {e[0]:.2e} rms {e[1]:.2e} {e[2] * 1e6:7.1f} us {2.0 * M * N * K / e[2] / 1e12:6.1f} TF"
    ok9 &= row[9][0] <= n0[0] and row[9][1] <= n0[1]
    ok6 &= row[6][0] <= n0[0] and row[6][1] <= n0[1]
    print(s, flush=True)
    lines.append(s)
ops.set_option("gemm_split_bf16", 0)
lines.append(f"# acceptance (max AND rms error of the split <= the native kernel's on every shape): n = 9: {'PASS' if ok9 else 'FAIL'}, n = 6: {'PASS' if ok6 else 'FAIL'}")
print(lines[-1])
out = ROOT / "gpurun_out" / "r06_gemm_split_bf16.txt"
out.parent.mkdir(exist_ok=True)
out.write_text("\n".join(lines) + "\n")
