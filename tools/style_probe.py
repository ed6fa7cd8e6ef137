#!/usr/bin/env python
"""The style encoder (attention type) forward + backward ALONE on the chip at the headline shape (B = 32, example length 384,
dropout on): the profiling target for its kernels' isolated durations (rocprofv3 --kernel-trace --stats; in the training
iteration they run beside the decoder's weight-gradient GEMMs, where a duration mostly measures the wait for a CU slot).
usage: [ZEGGS_OPTIONS=...] [STYLE_EVAL=1] python tools/style_probe.py [reps]      (STYLE_EVAL=1: eval mode, the kernels without dropout masks)"""
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd"), str(ROOT / "tests")]
from zeggs import modules, ops, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
torch.manual_seed(1234)
st = modules.StyleEncoder(synth.POSE_IN, 512, 64, type="attn", use_vae=True).to(dev)
st = st.eval() if os.environ.get("STYLE_EVAL") else st.train()
B, L = 32, 384
x = torch.randn(B, L, synth.POSE_IN, device=dev)
eps = torch.randn(B, 64, device=dev)
ops.manual_seed(5)


def step():
    st.zero_grad(set_to_none=True)
    z, mu, lv = st(x, 1.0, eps=eps)
    (z.sum() + 0.1 * mu.sum() + 0.1 * lv.sum()).backward()


step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    step()
torch.cuda.synchronize()
print(f"style encoder forward + backward alone, B = {B}, L = {L}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per pass")
