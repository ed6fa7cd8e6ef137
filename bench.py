#!/usr/bin/env python
"""Headline benchmark: gesture frames/sec of the ZeroEGGS training step on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is ONE full training iteration of BASELINE.json configs[1]: configs_v1 networks
(random-init, seed 1234), batch 32 per GPU of 256-frame windows cut from synthetic 60-fps 2-minute
clips that are resident in HBM, style examples of length 384: gather -> speech encoder -> style
VAE -> 255-step decoder rollout -> FK/L1 loss -> BPTT -> (RCCL all-reduce) -> fused RAdam.
Nothing is skipped or cached inside the timed region; training-mode dropout is on.
Prints ONE JSON line on rank 0 (contract: task statement / DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "ubisoft-laforge-zeroeggs_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

from zeggs import engine, modules, ops, synth  # noqa: E402

BATCH, WINDOW, EXAMPLE_LEN, CLIP_FRAMES = 32, 256, 384, 7200
H, SP, ST = 1024, 64, 64
XD = synth.POSE_IN + SP + ST
# algorithmic bytes of ONE decoder step, forward: every per-step weight + bias read once
STEP_WEIGHT_BYTES = 4 * (H * XD + 3 * H * (H + XD) + 3 * H * H + 3 * H * H + 3 * H * H + synth.POSE_OUT * H
                         + H + 4 * 3 * H + synth.POSE_OUT)
HBM_PEAK_GBS = 8000.0


def step_bytes(batch):
    return STEP_WEIGHT_BYTES + batch * 4 * (synth.POSE_IN + SP + ST + 2 * H + synth.POSE_OUT + 2 * H)


def build_dataset(n_train=64, n_unique=4, seed=0):
    """64 two-minute 60-fps clips (n_unique distinct, tiled: the content does not affect the timing)."""
    stats = synth.make_stats()
    base = [synth.make_clip(CLIP_FRAMES, seed=seed * 1000 + i, stats=stats) for i in range(n_unique)]
    clips = [base[i % n_unique] for i in range(n_train)]
    data = {k: np.concatenate([c[k] for c in clips], axis=0) for k in clips[0]}
    bounds = np.arange(n_train + 1) * CLIP_FRAMES
    data["ranges_train"] = np.stack([bounds[:-1], bounds[1:]], axis=1).astype(np.int32)
    data["ranges_train_labels"] = (np.arange(n_train) % 19).astype(np.int32)
    data.update(stats)
    return data


def build_nets(device):
    torch.manual_seed(1234)
    se = modules.SpeechEncoder(synth.N_AUDIO, 64, SP)
    de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, SP, ST, H, 2)
    st = modules.StyleEncoder(synth.POSE_IN, 512, ST, type="attn", use_vae=True)
    return se.to(device).train(), de.to(device).train(), st.to(device).train()


def cpu_baseline(data, iters=2):
    """The CPU oracle (torch-CPU restatement of the reference, oracle/) timed on this host's cores on the SAME
    workload shape (B=32, T=256, example 384): `iters` full iterations fwd+loss+bwd (+RAdam is negligible)."""
    from oracle import loss as oloss
    from oracle import nets as onets
    threads = min(64, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    torch.manual_seed(1234)
    se = modules.SpeechEncoder(synth.N_AUDIO, 64, SP)
    de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, SP, ST, H, 2)
    st = modules.StyleEncoder(synth.POSE_IN, 512, ST, type="attn", use_vae=True)
    ws = [{k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()} for m in (se, de, st)]
    B, T = BATCH, WINDOW
    f = lambda k: torch.as_tensor(np.stack([data[k][i * 100:i * 100 + T] for i in range(B)]))  # noqa: E731
    W = {k: f(k) for k in ("X_audio_features", "Y_root_pos", "Y_root_rot", "Y_root_vel", "Y_root_vrt", "Y_lpos",
                           "Y_ltxy", "Y_lvel", "Y_lvrt", "Y_gaze_pos")}
    t = lambda k: torch.as_tensor(np.asarray(data[k]), dtype=torch.float32)  # noqa: E731
    ex = torch.randn(B, EXAMPLE_LEN, synth.POSE_IN)
    eps = torch.randn(B, ST)
    times = []
    for it in range(iters + 1):
        t0 = time.perf_counter()
        speech = onets.speech_encoder(ws[0], (W["X_audio_features"] - t("audio_input_mean")) / t("audio_input_std"))
        z, mu, lv = onets.style_encoder(ws[2], ex, eps)
        O = onets.decoder_rollout(ws[1], W["Y_root_pos"][:, 0], W["Y_root_rot"][:, 0], W["Y_root_vel"][:, 0],
                                  W["Y_root_vrt"][:, 0], W["Y_lpos"][:, 0], W["Y_ltxy"][:, 0], W["Y_lvel"][:, 0],
                                  W["Y_lvrt"][:, 0], W["Y_gaze_pos"], speech, z.unsqueeze(1).repeat(1, T, 1),
                                  t("anim_input_mean"), t("anim_input_std"), t("anim_output_mean"),
                                  t("anim_output_std"), synth.DT)
        loss, _ = oloss.training_loss(O, [W[k] for k in ("Y_root_pos", "Y_root_rot", "Y_root_vel", "Y_root_vrt",
                                                         "Y_lpos", "Y_ltxy", "Y_lvel", "Y_lvrt")], W["Y_gaze_pos"],
                                      synth.PARENTS, synth.DT, mu, lv, iteration=it)
        loss.backward()
        for w in ws:
            for v in w.values():
                v.grad = None
        times.append(time.perf_counter() - t0)
    dt_ = float(np.mean(times[1:]))          # first iteration warms allocators / thread pools
    return {"value": round(B * T / dt_, 1), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{iters} full training iterations (B={B}, T={T}, example {EXAMPLE_LEN}) of the torch-CPU "
                      f"oracle, {dt_:.2f} s/iteration"}


def decode_rate(de, dev, T=1801):
    """second half of BASELINE.json's metric: autoregressive decode frames/s (config 5 regime: B=1, no_grad ring
    path, speech/style already encoded; 30 s of 60-fps frames), measured after the timed training steps."""
    st = {k: torch.as_tensor(v, dtype=torch.float32, device=dev) for k, v in synth.make_stats().items() if k.startswith("anim")}
    g = torch.Generator(device="cpu").manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    args = (de.eval(), r(1, synth.POSE_OUT), torch.zeros(1, 3, device=dev), torch.tensor([[1.0, 0, 0, 0]], device=dev),
            r(1, T, 3) * 10, r(1, T, SP) * 0.3, r(1, T, ST) * 0.3, st["anim_input_mean"], st["anim_input_std"],
            st["anim_output_mean"], st["anim_output_std"], synth.DT)
    with torch.no_grad():
        ops.decoder_core(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.decoder_core(*args)
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
    de.train()
    return {"value": round((T - 1) / dt_, 1), "unit": "frames/s", "us_per_frame": round(dt_ * 1e6 / (T - 1), 2),
            "config": f"B=1 autoregressive rollout of {T - 1} frames, no_grad ring path (GEMV stage kernels)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world,
                                             device_id=torch.device("cuda", local))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    data = build_dataset()
    ds = engine.DeviceDataset(data, WINDOW, dev)
    se, de, st = build_nets(dev)
    eng = engine.TrainEngine(se, de, st, ds, synth.PARENTS, synth.DT, world_size=world, rank=rank)
    ops.manual_seed(1000 + rank)
    perm_rng = np.random.default_rng(42)                    # same permutation on every rank
    perm = perm_rng.permutation(len(ds))
    gb = BATCH * world

    def indices(it):
        return engine.shard_indices(perm, it % (len(ds) // gb), BATCH, world, rank)

    torch.manual_seed(77 + rank)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for it in range(a.warmup):
        eng.step(indices(it), EXAMPLE_LEN)
    sync()
    eng.decoder_fwd_events = []          # HIP events (torch's current stream = the stream the kernels run on)
    sync()
    t0 = time.perf_counter()
    for it in range(a.warmup, a.warmup + a.steps):
        eng.step(indices(it), EXAMPLE_LEN)
    sync()
    el = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(el)
    loss = float(eng.step(indices(a.warmup + a.steps), EXAMPLE_LEN))
    if rank == 0:
        ms = elapsed / a.steps * 1e3
        out = {
            "metric": "train_frames_per_sec", "value": round(gb * WINDOW * a.steps / elapsed, 1), "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs_v1.json nets (25.5M params, random-init), batch 32/GPU x 256-frame windows "
                                   "from synthetic 60fps 2-min clips, style example 384 frames, full train step "
                                   "(gather+fwd+loss+bwd+allreduce+RAdam)",
                       "global_batch": gb, "window": WINDOW, "parallelism": f"dp{world}"},
            "final_loss": round(loss, 4),
        }
        ev = eng.decoder_fwd_events[:a.steps]
        probe = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev])) * 1e-3 / (WINDOW - 1)   # s per decoder step
        ach = step_bytes(BATCH) / probe / 1e9
        pmc_file = ROOT / "profiles" / "r01_decoder_step_pmc.json"
        traffic = json.load(open(pmc_file))["traffic_bytes_per_step"] if pmc_file.exists() else None
        out["roofline"] = {"bound": "hbm", "kernel": "decoder forward step = 3 launches of stage_k (GRU l0, GRU l1, "
                                                     "layer2+pose integration+next layer0), per-step figures",
                           "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "us_per_step": round(probe * 1e6, 2), "algorithmic_bytes_per_step": step_bytes(BATCH),
                           "note": "HIP events around the 255-step forward rollout inside the timed iterations; "
                                   "traffic = FETCH_SIZE(x2)+WRITE_SIZE from profiles/r01_decoder_step_pmc.json; at "
                                   "B=32 the step is also at the fp32 MFMA ridge (1.24 GFLOP/step)"}
        if world == 1:
            out["decode"] = decode_rate(de, dev)
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(data)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
