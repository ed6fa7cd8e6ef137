"""mel front-end A/B: the FFT form (round 4, default) against the fp64-MFMA DFT (round 3) and the direct-DFT kernel -- time for 30
minutes of audio (wav resident on the device) and agreement of the features."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
from zeggs import audio, ops  # noqa: E402

n = 30 * 60 * 16000
wav = torch.as_tensor((0.1 * np.random.default_rng(0).standard_normal(n)).astype(np.float32)).cuda()
T = audio.n_anim_frames(n)
outs = {}
for name, fft, mfma in (("direct", 0, 0), ("mfma", 0, 1), ("fft", 1, 1)):
    ops.set_option("mel_fft", fft)
    ops.set_option("mel_mfma", mfma)
    audio.mel_features(wav[:16000], 60)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        o = audio.mel_features(wav, T)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    outs[name] = o.cpu()
    print(f"{name:7s}: {min(ts) * 1e3:8.3f} ms for {T} frames (wav on the device)", flush=True)
ops.set_option("mel_fft", 1)
for a, b in (("direct", "mfma"), ("direct", "fft"), ("mfma", "fft")):
    d = (outs[a] - outs[b]).abs()
    print(f"max |{a} - {b}|: {float(d[torch.isfinite(d)].max()):.3e}   NaN pattern equal: "
          f"{bool((torch.isnan(outs[a]) == torch.isnan(outs[b])).all())}")
# round 5: the log chain (option mel_exact_log: 0 hardware log2 / exp2 (default), 2 float64 affine, 1 literal float64 chain)
for mode in (1, 2, 0):
    ops.set_option("mel_exact_log", mode)
    audio.mel_features(wav[:16000], 60)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        o = audio.mel_features(wav, T)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    outs[f"log{mode}"] = o.cpu()
    print(f"fft, mel_exact_log={mode}: {min(ts) * 1e3:8.3f} ms", flush=True)
for m in (2, 0):
    d = (outs[f"log{m}"] - outs["log1"]).abs()
    print(f"max |mode {m} - literal chain|: {float(d[torch.isfinite(d)].max()):.3e}")
