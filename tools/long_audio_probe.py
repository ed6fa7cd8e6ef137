#!/usr/bin/env python
"""BASELINE.json configs[4] shape: 30 minutes of 16 kHz audio through the device front-end + speech encoder + B=1 decode
(random-init nets, synthetic audio): stage timings and finiteness."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
from zeggs import audio, modules, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
n = int(minutes * 60 * 16000)
rng = np.random.default_rng(0)
wav = (0.1 * rng.standard_normal(n)).astype(np.float32)
torch.manual_seed(0)
se = modules.SpeechEncoder(synth.N_AUDIO, 64, 64).to(dev).eval()
de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, 64, 64, 1024, 2).to(dev).eval()
st = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32, device=dev) for k, v in synth.make_stats().items()}


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


T = audio.n_anim_frames(n)
with torch.no_grad():
    audio.mel_features(wav[:16000], 60)                                                   # warm-up
    feats, t_mel = timed(lambda: audio.mel_features(wav, T))
    x = ((feats[None] - st["audio_input_mean"]) / st["audio_input_std"]).contiguous()
    se(x[:, :512].contiguous())                                                            # warm-up
    sp, t_se = timed(lambda: se(x))
    args = (de, torch.randn(1, synth.POSE_OUT, device=dev), torch.zeros(1, 3, device=dev),
            torch.tensor([[1.0, 0, 0, 0]], device=dev), torch.randn(1, T, 3, device=dev), sp,
            torch.randn(1, T, 64, device=dev) * 0.3, st["anim_input_mean"], st["anim_input_std"], st["anim_output_mean"],
            st["anim_output_std"], synth.DT)
    out, t_dec = timed(lambda: ops.decoder_core(*args))
print(f"{minutes:g} min of audio = {T} frames: mel {t_mel * 1e3:.1f} ms, speech encoder {t_se * 1e3:.1f} ms, "
      f"decode {t_dec:.2f} s ({T / t_dec:.0f} frames/s, {T / 60.0 / (t_mel + t_se + t_dec):.0f}x real time); "
      f"finite: {bool(torch.isfinite(out[0]).all() and torch.isfinite(feats).all())}")
