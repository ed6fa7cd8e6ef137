#!/usr/bin/env python
"""Round 5: what an event record costs behind a kernel that leaves the L2s dirty (the iteration timeline shows ~65 us of an idle chip
between the optimizer's last kernel and the next iteration's first).  Loop of [big write kernel, <variant>, small kernel] timed by
wall clock; the difference to the bare loop is the price of the variant."""
import time

import torch

dev = torch.device("cuda:0")
x = torch.zeros(75_000_000, device=dev)          # 300 MB: an RAdam-sized write
y = torch.zeros(1024, device=dev)
side = torch.cuda.Stream()
N = 60


def loop(variant):
    evs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        x.add_(1.0)
        if variant == "timing_event":
            e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
        elif variant == "plain_event":
            e = torch.cuda.Event(); e.record(); evs.append(e)
        elif variant == "event_waited_by_side_stream":
            e = torch.cuda.Event(); e.record(); side.wait_event(e); evs.append(e)
        elif variant == "d2h_copy_on_side_stream":
            e = torch.cuda.Event(); e.record()
            with torch.cuda.stream(side):
                side.wait_event(e)
                pin.copy_(y[:4], non_blocking=True)
        elif variant == "d2h_copy_same_stream":
            pin.copy_(y[:4], non_blocking=True)
        y.add_(1.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6


pin = torch.zeros(4).pin_memory()
for v in ("bare", "timing_event", "plain_event", "event_waited_by_side_stream", "d2h_copy_on_side_stream", "d2h_copy_same_stream", "bare"):
    loop(v)
    print(f"{v:32s} {loop(v):8.1f} us per round", flush=True)
