"""The streaming BVH writer of generate_gesture() on 108 000 frames, alone, in variants: what does the host side underneath the
chunked persistent decode cost the DECODE (the one-launch rollout takes 1.04 s)?  usage: stream_probe.py"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd")]
import bench  # noqa: E402
from zeggs import anim, generate, modules, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1234)
de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, 64, 64, 1024, 2).to(dev).eval()
T = 108000
args = bench.decode_args(de, dev, T)
_, pose0, rpos0, rrot0, gaze, speech, style, im, isd, om, osd, dt = args
J = len(synth.PARENTS)
names = [f"j{i}" for i in range(J)]
tmp = Path(tempfile.mkdtemp(prefix="zeggs_sp_"))


def run(label, **kw):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        generate._decode_to_bvh_streaming(de, pose0, rpos0, rrot0, gaze[0, :1], speech, style, (im, isd, om, osd), dt,
                                          str(tmp / "o.bvh"), synth.PARENTS, names, **kw)
    torch.cuda.synchronize()
    print(f"{label:46s} {(time.perf_counter() - t0) * 1e3:8.1f} ms  {generate.PROFILE}", flush=True)


with torch.no_grad():
    ops.decoder_core(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops.decoder_core(*args)
    torch.cuda.synchronize()
    print(f"{'one launch, frames kept on the device':46s} {(time.perf_counter() - t0) * 1e3:8.1f} ms")
    t0 = time.perf_counter()
    state, k = (pose0, rpos0, rrot0, None), 0
    while k < T - 1:
        n = min(8192, T - 1 - k)
        p, rp, rr, h = ops.decoder_chunk(de, state[0], state[1], state[2], gaze[:, k:k + n + 1], speech[:, k:k + n + 1],
                                         style[:, k:k + n + 1], im, isd, om, osd, dt, h_in=state[3])
        state, k = (p[:, -1], rp[:, -1], rr[:, -1], h), k + n
    torch.cuda.synchronize()
    print(f"{'chunks of 8192, frames dropped':46s} {(time.perf_counter() - t0) * 1e3:8.1f} ms")
generate.PROFILE = {}
run("streaming writer (warm-up)")
run("streaming writer")
run("streaming writer, 1 formatting thread", threads=1)
fmt = anim.format_rows
anim.format_rows = lambda a: b""
run("... rows not formatted (copies only)")
anim.format_rows = fmt
run("streaming writer, chunk 16384", chunk=16384)
run("streaming writer, chunk 32768", chunk=32768)
