#!/usr/bin/env python
"""The style encoder's first convolution (43 GFLOP) as it runs today -- the padded input [B (L + 2)][C] read as a [B (L + 2) - 2, 3 C]
matrix with overlapping rows, LDS-tiled stream-K (csrc/encoders.hip: conv_flat) -- against the same product on the barrier-free direct TN
kernel: the input stored TRANSPOSED ([C][B (L + 2)], k-major), the three taps as three batch-reduce segments (kbatch = 3: segment s
reads the input shifted by s columns and the s-th tap's weight block).  usage: python tools/conv0_tn_probe.py [shield]"""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
from zeggs import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
B, L, Cc, H = 32, 384, synth.POSE_IN, 512
LP, M = L + 2, 32 * 386 - 2
L_ = ops.lib()
if len(sys.argv) > 1:
    L_.zeggs_gemm_route(1, 1, 8, 0)
xp = torch.randn(B * LP, Cc, device=dev)
xpT = xp.t().contiguous()                       # [C][B LP]
Wf = torch.randn(3 * Cc, H, device=dev)         # k-major packed weights: row tap * C + c
c_nn = torch.zeros(B * LP, H, device=dev)
c_tn = torch.zeros(B * LP, H, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731


def nn():
    rc = L_.zeggs_gemm(P(xp), P(Wf), P(c_nn), None, M, H, 3 * Cc, C.c_long(Cc), C.c_long(1), C.c_long(H), C.c_long(1), C.c_long(H),
                       C.c_long(1), 1, C.c_long(0), C.c_long(0), C.c_long(0), C.c_float(1.0), C.c_float(0.0), 0, S())
    assert rc == 0, L_.zeggs_last_error()


def tn():
    rc = L_.zeggs_gemm_kbatch(P(xpT), P(Wf), P(c_tn), M, H, Cc, C.c_long(1), C.c_long(B * LP), C.c_long(H), C.c_long(1), C.c_long(H),
                              C.c_long(1), 3, C.c_long(1), C.c_long(Cc * H), C.c_float(0.0), S())
    assert rc == 0, L_.zeggs_last_error()


L_.zeggs_last_error.restype = C.c_char_p
for name, f in (("NN stream-K (today)", nn), ("TN kbatch=3 direct", tn)):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e2
    print(f"{name:22s} {us:7.1f} us  {2.0 * M * H * 3 * Cc / us / 1e6:6.1f} TFLOP/s")
ref = (xp[:64].double() @ Wf[:Cc].double() + xp[1:65].double() @ Wf[Cc:2 * Cc].double() + xp[2:66].double() @ Wf[2 * Cc:].double())
print("max |NN - f64|", float((c_nn[:64].double() - ref).abs().max()), " max |TN - f64|", float((c_tn[:64].double() - ref).abs().max()),
      " max |NN - TN|", float((c_nn[:M] - c_tn[:M]).abs().max()))
