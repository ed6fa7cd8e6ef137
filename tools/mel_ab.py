import sys, time
sys.path[:0] = ["/root/repo", "/root/repo/ubisoft-laforge-zeroeggs_amd"]
import numpy as np, torch
from zeggs import audio, ops
n = 30 * 60 * 16000
wav = (0.1 * np.random.default_rng(0).standard_normal(n)).astype(np.float32)
T = audio.n_anim_frames(n)
outs = {}
for v in (0, 1):
    ops.set_option("mel_mfma", v)
    audio.mel_features(wav[:16000], 60)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); o = audio.mel_features(wav, T); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    outs[v] = o.cpu()
    print(f"mel_mfma={v}: {min(ts)*1e3:.2f} ms (incl. H2D of the wav)", flush=True)
d = (outs[0] - outs[1]).abs()
print("max |direct - mfma|:", float(d[torch.isfinite(d)].max()), "nan pattern equal:", bool((torch.isnan(outs[0]) == torch.isnan(outs[1])).all()))
