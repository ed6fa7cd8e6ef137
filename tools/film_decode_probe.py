"""B = 1 free-running decode with the FiLM decoder (GEMV stage launches; the persistent kernel is rnn_cond = "normal" only):
microseconds per frame, stage path vs the generic per-step path.  usage: film_decode_probe.py [frames]"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
import bench  # noqa: E402
from zeggs import modules, ops, synth  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda:0")
torch.manual_seed(1234)
de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, 64, 64, 1024, 2, rnn_cond="film").to(dev).eval()
args = bench.decode_args(de, dev, T)
for label, fast in (("stage launches (4 per frame)", 1), ("generic per-step path", 0)):
    ops.set_option("decoder_fast", fast)
    with torch.no_grad():
        ops.decoder_core(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.decoder_core(*args)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"FiLM B = 1, {T} frames, {label:30s}: {dt / (T - 1) * 1e6:6.2f} us per frame ({(T - 1) / dt:8.0f} frames/s)")
ops.set_option("decoder_fast", 1)
