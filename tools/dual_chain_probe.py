#!/usr/bin/env python
"""Would two independent half-batch recurrences on two streams beat one full-batch chain?  Times the decoder
rollout (no_grad and forward+backward) as ONE B=32 call vs TWO concurrent B=16 calls (2 host threads, 2 streams)."""
import sys
import threading
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
from zeggs import modules, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, 64, 64, 1024, 2).to(dev)
s = {k: torch.as_tensor(v, dtype=torch.float32, device=dev) for k, v in synth.make_stats().items() if k.startswith("anim")}
T = 256


def make(B, grad):
    sp = torch.randn(B, T, 64, device=dev) * 0.3
    if grad:
        sp.requires_grad_(True)
    return (de, torch.randn(B, synth.POSE_OUT, device=dev), torch.zeros(B, 3, device=dev),
            torch.tensor([[1.0, 0, 0, 0]], device=dev).repeat(B, 1), torch.randn(B, T, 3, device=dev) * 10, sp,
            torch.randn(B, T, 64, device=dev) * 0.3, s["anim_input_mean"], s["anim_input_std"], s["anim_output_mean"],
            s["anim_output_std"], synth.DT)


def run(args, grad, stream=None):
    with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream()):
        if grad:
            p, a, b = ops.decoder_core(*args)
            (p.sum() + a.sum() + b.sum()).backward()
        else:
            with torch.no_grad():
                ops.decoder_core(*args)


def timed(fn, reps=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for grad in (False, True):
    full, h0, h1 = make(32, grad), make(16, grad), make(16, grad)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        th = [threading.Thread(target=run, args=(h0, grad, s0)), threading.Thread(target=run, args=(h1, grad, s1))]
        [t.start() for t in th]
        [t.join() for t in th]

    one = timed(lambda: run(full, grad))
    half = timed(lambda: run(h0, grad))
    two = timed(both)
    print(f"{'fwd+bwd' if grad else 'fwd    '}: one B=32 call {one:.2f} ms | one B=16 call {half:.2f} ms | "
          f"two concurrent B=16 calls {two:.2f} ms", flush=True)
