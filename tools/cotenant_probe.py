#!/usr/bin/env python
"""Multi-GPU readiness measured on ONE GPU: a stand-in collective beside the iteration's tail.

The data-parallel schedule of zeggs.engine.TrainEngine (give-up flag first, the decoder's gradient halves exchanged underneath
the weight-gradient products / the encoders' backward, early RAdam slices behind them) runs on a single rank with
`torch.distributed.all_reduce` replaced by `zeggs_test_cotenant`: W workgroups resident on their own stream for a stretch of wall
clock per exchanged slice (proportional to its bytes at an assumed bus rate) -- what RCCL's channel workgroups are to the other
kernels on the chip.  Measured: ms per iteration for `gemm_direct_reserve` in {0, 16, 32, 48} x W in {32, 48}; the reserve is
what the shielded stream-K products leave out of their grids so that their equal-share workgroups are all resident.

    python tools/cotenant_probe.py [steps]        -> profiles/r06_reserve_ab.txt (one line per setting)"""
import ctypes as C
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # as bench.launch_env gives every rank: the engine's three streams + the collective's must not share a hardware queue

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "ubisoft-laforge-zeroeggs_amd"), str(ROOT / "tests")]
import bench  # noqa: E402
from zeggs import engine, ops  # noqa: E402

dev = torch.device("cuda:0")
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
BUS_GBS = float(os.environ.get("BUS_GBS", 300.0))          # assumed bus bandwidth of an 8-rank ring all-reduce over xGMI (RCCL, large messages)
GRID = [(int(w), int(r)) for w, r in (x.split(":") for x in os.environ.get("GRID", "32:0,32:16,32:32,32:48,48:0,48:16,48:32,48:48").split(","))]
COMM = torch.cuda.Stream(priority=-1) if int(os.environ.get("COMM_HIGH", "0")) else torch.cuda.Stream()
scratch = torch.zeros(1 << 18, device=dev)
STATE = {"W": 32, "windows_ms": []}


class FakeWork:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def fake_all_reduce(tensor, op=None, group=None, async_op=False):
    """ordered behind the kernels already on the current stream (as a process group's collective is), resident for
    2 (n - 1) / n x bytes / bus rate (n = 8) + 20 us"""
    ms = 1.75 * tensor.numel() * 4 / (BUS_GBS * 1e9) * 1e3 + 0.02
    STATE["windows_ms"].append(ms)
    comm = STATE.get("comm") or COMM      # COMM_REUSE=1: the engine's third stream instead of a stream of its own (stream-count A/B)
    comm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(comm):
        if STATE["W"] > 0:      # (W = 0: the exchange's stream dependencies without anything resident -- what the ORDER alone costs)
            ops._check(ops.lib().zeggs_test_cotenant(STATE["W"], 256, C.c_float(ms), C.c_void_p(scratch.data_ptr()),
                                                     C.c_long(scratch.numel()), C.c_void_p(comm.cuda_stream)), "test_cotenant")
        ev = torch.cuda.Event()
        ev.record(comm)
    w = FakeWork(ev)
    if not async_op:
        w.wait()
    return w


def run(reserve, W, fake=True):
    STATE["W"], STATE["windows_ms"] = W, []
    data = bench.build_dataset()
    ds = engine.DeviceDataset(data, bench.WINDOW, dev)
    se, de, st = bench.build_nets(dev)
    real = torch.distributed.all_reduce
    if fake:
        torch.distributed.all_reduce = fake_all_reduce
    try:
        eng = engine.TrainEngine(se, de, st, ds, bench.synth.PARENTS, bench.synth.DT, force_allreduce=fake,
                                 overlap_allreduce=bool(int(os.environ.get("OVERLAP", "1"))),
                                 early_decoder_step=bool(int(os.environ.get("EARLY", "1"))))
        eng.ctx.gemm_route = (1, 1, 8, int(reserve))
        STATE["comm"] = eng.aux_stream if int(os.environ.get("COMM_REUSE", "0")) else None
        perm = np.random.default_rng(0).permutation(len(ds))
        idx = lambda it: engine.shard_indices(perm, it % (len(ds) // bench.BATCH), bench.BATCH, 1, 0)  # noqa: E731
        for it in range(6):
            eng.step(idx(it), bench.EXAMPLE_LEN)
        per = []
        for r in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for it in range(STEPS):
                k = 6 + r * STEPS + it
                h0 = time.perf_counter()
                eng.step(idx(k), bench.EXAMPLE_LEN)
                STATE["host_ms"] = STATE.get("host_ms", 0.0) + (time.perf_counter() - h0) * 1e3 / (3 * STEPS)
                if it + 1 < STEPS:
                    eng.prefetch(idx(k + 1), bench.EXAMPLE_LEN)      # (the next batch's gather on the third stream, as bench.py's loop)
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) / STEPS * 1e3)
        print(f"   (host time inside eng.step: {STATE.pop('host_ms', 0.0):.2f} ms per iteration)", flush=True)
        nwin = len(STATE["windows_ms"]) // max(1, 6 + 3 * STEPS)
        win = STATE["windows_ms"][-nwin:] if nwin else []
        del eng
        return per, win
    finally:
        torch.distributed.all_reduce = real


lines = [f"# tools/cotenant_probe.py {STEPS}: ms per iteration (three regions of {STEPS} steps), B = 32 x 256, one MI355X; stand-in collective of W x 256 "
         f"threads resident per exchanged slice for 1.75 x bytes / {BUS_GBS:g} GB/s + 20 us on its own stream"]
per = [0.0]
if not int(os.environ.get("SKIP_BASE", "1" if len(GRID) == 1 else "0")):
    per, _ = run(0, 32, fake=False)
lines.append(f"no exchange (single-rank schedule), reserve 0                       {np.mean(per):7.3f}   regions {[round(x, 3) for x in per]}")
print(lines[-1], flush=True)
for W, reserve in GRID:
    if True:
        per, win = run(reserve, W)
        lines.append(f"stand-in collective W = {W:2d}, gemm_direct_reserve = {reserve:2d}              {np.mean(per):7.3f}   regions {[round(x, 3) for x in per]}"
                     f"   windows per iteration (ms) {[round(x, 3) for x in win]}")
        print(lines[-1], flush=True)
out = ROOT / "gpurun_out" / os.environ.get("OUT", "r06_reserve_ab.txt")
out.parent.mkdir(exist_ok=True)
out.write_text("\n".join(lines) + "\n")
